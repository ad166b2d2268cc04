#!/usr/bin/env python
"""Replay volume of the block-wise exact float32 chain (tests/exact_chain_proto.py) at BASELINE config 3's and config 4's
geometry, on a sample of the columns (CPU, numpy): how many (block, column) pairs a scan has to replay sequentially, by
cause, on one GPU and per rank of an 8-rank row-sharded job whose blocks take their binades from the all-gathered float64
totals of the earlier ranks.  Checks the result against the plain sequential chain on the way.
    python tools/exact_chain_stats.py [rows] [columns]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import exact_chain_proto as P  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 128


def seq_chain(X):
    s = np.zeros(X.shape[1], dtype=np.float32)
    for r in range(X.shape[0]):
        s = (s + X[r]).astype(np.float32)
    return s


def report(name, st, n_rows, block):
    b = max(st["blocks"], 1)
    print(f"  {name:44s} non-empty (block, column) pairs {st['blocks']:9d}  replayed {st['replayed']:7d} = "
          f"{100.0 * st['replayed'] / b:6.3f} %   [no start yet {st['no_start']}, crossing {st['crossing']}, "
          f"estimate's binade wrong {st['estimate']}, tie / sign {st['tie_or_sign']}]   replayed rows per column "
          f"{st['replayed'] * block / cols:8.1f} of {n_rows}", flush=True)


def run(label, X, block):
    t0 = time.time()
    want = seq_chain(X)
    print(f"{label}: {X.shape[0]} rows x {X.shape[1]} sampled columns, blocks of {block} rows "
          f"(sequential chain: {time.time() - t0:.0f} s)", flush=True)
    st = {}
    got = P.chain_by_blocks(X, block=block, stats=st)
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    report("one GPU (start = 0)", st, X.shape[0], block)
    # 8 ranks: records from the all-gathered float64 totals (concurrent), scan from the exact hand-over (in rank order)
    R = 8
    bounds = [(k * X.shape[0] // R // block * block, (k + 1) * X.shape[0] // R // block * block if k < R - 1 else X.shape[0])
              for k in range(R)]
    totals = [X[a:b].sum(axis=0, dtype=np.float64) for a, b in bounds]
    s = np.zeros(X.shape[1], dtype=np.float32)
    for k, (a, b) in enumerate(bounds):
        est = np.sum(totals[:k], axis=0) if k else np.zeros(X.shape[1])
        recs = P.rank_records(X[a:b], est, block=block)
        st = {}
        s = P.rank_scan(X[a:b], s, recs, block=block, stats=st)
        report(f"rank {k} of {R} (rows {a} .. {b})", st, b - a, block)
    assert np.array_equal(s.view(np.int32), want.view(np.int32))
    print("  -> the 8-rank form equals the sequential chain bit for bit", flush=True)


rs = np.random.RandomState(2)
# config 2 / 3: gamma(0.3, 1) with entries < 0.5 set to 0 (~19 % stored)
X = rs.gamma(0.3, 1.0, (rows, cols)).astype(np.float32)
X[X < 0.5] = 0
for block in (64, 256):
    run("config 3 (dense fp32, gamma(0.3) >= 0.5)", X, block)
# config 4: 7 % Bernoulli mask x log1p(1 + floor(rand^3 * 8)) -- eight distinct values -- scaled by fl32(1 / n) as scipy does
half = rows // 2
M = rs.rand(half, cols) < 0.07
V = np.log1p(1.0 + np.floor(rs.rand(half, cols) ** 3 * 8.0)).astype(np.float32)
Y = np.where(M, (V * np.float32(1.0 / half)).astype(np.float32), np.float32(0)).astype(np.float32)
run("config 4 (CSR 7 %, 8 distinct values x fl32(1/n))", Y, 64)
# real-data-like CSR: continuous values (log1p of normalised counts)
V2 = np.log1p(rs.gamma(0.5, 2.0, (half, cols))).astype(np.float32)
Y2 = np.where(M, (V2 * np.float32(1.0 / half)).astype(np.float32), np.float32(0)).astype(np.float32)
run("CSR 7 %, continuous values x fl32(1/n)", Y2, 64)

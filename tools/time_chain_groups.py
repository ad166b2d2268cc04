#!/usr/bin/env python
"""Time of one rank's reference-order chain pass (icv_colchain) over its shard when the columns are taken in T groups
(dist.reference_means_chained pipelines the ranks over these groups): config 3's per-rank shards, dense fp32 x 20 000 genes.
    python tools/time_chain_groups.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from infercnvpy_amd import _engine, dist as icd  # noqa: E402

G = 20000
for rows in (125_000, 250_000, 500_000):
    X = torch.rand((rows, G), device="cuda", dtype=torch.float32)
    dm = _engine.DeviceMatrix(dense=X)
    for T in (1, 2, 4, 8, 16):
        groups = icd.chain_column_groups(G, 4, T)
        acc = torch.zeros(G, dtype=torch.float32, device="cuda")
        for _ in range(2):
            for c in groups:
                _engine.column_chain(dm, acc, None, rows, cols=None if T == 1 else c)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        reps = 5
        for _ in range(reps):
            for c in groups:
                _engine.column_chain(dm, acc, None, rows, cols=None if T == 1 else c)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        print(f"rows {rows:7d}  T {T:2d}  whole pass {ms:7.3f} ms  per group {ms / T:7.3f} ms  "
              f"({rows * G * 4 / ms / 1e6:6.0f} GB/s over the pass)", flush=True)
    del X, dm
    torch.cuda.empty_cache()

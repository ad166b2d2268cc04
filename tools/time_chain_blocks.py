#!/usr/bin/env python
"""One rank's share of the reference means by integer blocks (dist.reference_means_blocks) on one GPU: the three calls on
config 3's per-rank shard (125 000 x 20 000 dense fp32 at 8 ranks), continuing a running chain, against the chain
kernel's pass over the same rows.   python tools/time_chain_blocks.py [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from infercnvpy_amd import _engine  # noqa: E402

G = 20000
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


A = bench.synth_rows(torch, 0, rows, G)
B = bench.synth_rows(torch, rows, 2 * rows, G)
dmA, dmB = _engine.DeviceMatrix(dense=A), _engine.DeviceMatrix(dense=B)
sA = _engine.column_chain(dmA, None, None, 2 * rows)
want = _engine.column_chain(dmB, sA.clone(), None, 2 * rows)
cbA, cbB = _engine.ChainBlocks(dmA), _engine.ChainBlocks(dmB)
tA = cbA.sums()
ms_chain = timed(lambda: _engine.column_chain(dmB, sA.clone(), None, 2 * rows))
ms_sums = timed(lambda: cbB.sums())
ms_rec = timed(lambda: cbB.records(tA))
cbB.replayed.zero_()
got = cbB.scan(sA.clone())
rep = int(cbB.replayed.item())
ms_scan = timed(lambda: cbB.scan(sA.clone()))
assert torch.equal(got.view(torch.int32), want.view(torch.int32))
gb = rows * G * 4 / 1e9
print(f"shard {rows} x {G} fp32 ({gb:.1f} GB), continuing the chain of the {rows} rows before it; result = k_colchain's bits")
print(f"  k_colchain (the chained form's turn of this rank)        {ms_chain:7.3f} ms  ({gb / ms_chain:6.2f} TB/s)")
print(f"  blocks 1. float64 totals (k_colsum_dense, concurrent)     {ms_sums:7.3f} ms  ({gb / ms_sums:6.2f} TB/s)")
print(f"  blocks 2. block records  (k_chain_records, concurrent)    {ms_rec:7.3f} ms  ({gb / ms_rec:6.2f} TB/s)")
print(f"  blocks 3. scan           (k_chain_scan, in rank order)    {ms_scan:7.3f} ms  replayed {rep} of {cbB.n_blocks()} "
      f"(block, column) pairs = {100.0 * rep / cbB.n_blocks():.3f} %")
cbA.records(None)
cbA.replayed.zero_()
g0 = cbA.scan(torch.zeros(G, dtype=torch.float32, device="cuda"))
assert torch.equal(g0.view(torch.int32), sA.view(torch.int32))
rep0 = int(cbA.replayed.item())
ms_scan0 = timed(lambda: cbA.scan(torch.zeros(G, dtype=torch.float32, device="cuda")))
print(f"  first rank (chains start at 0): scan {ms_scan0:7.3f} ms, replayed {100.0 * rep0 / cbA.n_blocks():.3f} %")
R = 8
print(f"  => {R} ranks: means = totals + records (concurrent) + {R} scans = {ms_sums + ms_rec + ms_scan0 + (R - 1) * ms_scan:6.2f} ms "
      f"(pipelined over 4 column groups: ~{ms_sums + ms_rec + (ms_scan0 + (R - 1) * ms_scan) * (R + 3) / (4 * R):5.2f} ms) "
      f"against {(2 + R - 1) / 2 * ms_chain:6.2f} ms for the chained form pipelined over 2 groups")

#!/usr/bin/env python
"""Threshold + CSR pack of config 2's x_res in ONE pass pair over all rows against the same in pieces of
whole chunks: does the second read of x_res (k_csr_fill_ring after k_thr_mask_ring) come cheaper when the piece it
re-reads is small enough to still sit in the 256 MB memory-side cache?  Timing experiment only (the pieces pack into
separate buffers).     python tools/time_pack_pieces.py [cells]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import bench  # noqa: E402
import cases  # noqa: E402
from infercnvpy_amd import _engine  # noqa: E402
from infercnvpy_amd._plan import GenePlan  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
CS = 5000
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
X = bench.synth_rows(torch, 0, cells, bench.G)
dm = _engine.to_device_matrix(X, torch.float32)
ref = (_engine.column_sums(dm)[0] / cells).float()
res = _engine.run_hot_path(plan, dm, ref, None, chunksize=CS, apply=False)
torch.cuda.synchronize()
W = plan.n_windows


def piece(r0, r1):
    sub = _engine.SmoothResult(res.out[r0:r1], res.cell_median[r0:r1], None, res.thr[r0 // CS:(r1 + CS - 1) // CS], None)
    return _engine.threshold_csr(plan, dm, ref, None, sub, lfc_clip=3.0, chunksize=CS, row0=r0, row1=r1)


def timed(rows_per_piece, reps=30):
    def once():
        return [piece(r, min(cells, r + rows_per_piece)) for r in range(0, cells, rows_per_piece)]
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pk = once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, sum(int(p.nnz()) for p in pk)


print(f"{cells} cells x {W} windows float32 = {cells * W * 4 / 1e6:.0f} MB of x_res; threshold + pack per pass over all rows")
for rpp in (cells, 50_000, 25_000, 10_000, 5000):
    if rpp > cells:
        continue
    ms, nnz = timed(rpp)
    print(f"  pieces of {rpp:7d} rows ({rpp * W * 4 / 1e6:6.1f} MB each): {ms:7.3f} ms   kept {nnz}")

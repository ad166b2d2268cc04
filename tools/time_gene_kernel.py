#!/usr/bin/env python
"""k_gene_fused alone (icv_gene_values_from_windows) at config 2's geometry: float64 windows of 100 000 cells in, the
cells x genes float64 layer out; events on the launch stream.   python tools/time_gene_kernel.py [cells]"""
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
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
X = bench.synth_rows(torch, 0, cells, bench.G)
dm = _engine.to_device_matrix(X, torch.float32)
ref = (_engine.column_sums(dm)[0] / cells).float()
res = _engine.run_hot_path(plan, dm, ref, None, chunksize=5000, windows=True)
out = torch.empty((cells, bench.G), dtype=torch.float64, device="cuda")
W = plan.n_windows


def once():
    _engine.gene_values_from_windows(plan, res.windows, thr=res.thr, chunksize=5000, n_vars=bench.G, out=out)


for _ in range(3):
    once()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    once()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
b = cells * (8 * W + 8 * bench.G)
print(f"k_gene_fused {cells} cells x {W} windows -> {bench.G} genes: median {ts[5]:.3f} ms (min {ts[0]:.3f}) = "
      f"{b / ts[5] / 1e9:.2f} TB/s of {b / 1e9:.2f} GB")

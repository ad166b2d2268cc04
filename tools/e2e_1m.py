#!/usr/bin/env python
"""north_star's size from HOST memory on one GPU: tl.infercnv on a 1 000 000 x 20 000 dense fp32 matrix (80 GB of host
memory; generated on the GPU chunk by chunk, config 2's generator), reference = all-cell mean (the default call) and a
given reference.  python tools/e2e_1m.py [cells]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import cases  # noqa: E402
import infercnvpy_amd as cnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = 20000
X = np.empty((n, G), dtype=np.float32)
for r0 in range(0, n, 50_000):
    r1 = min(n, r0 + 50_000)
    X[r0:r1] = bench.synth_rows(torch, r0, r1, G).cpu().numpy()
torch.cuda.empty_cache()
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
ref = np.asarray(X[:2000].mean(axis=0), dtype=np.float64).astype(np.float32)
for rep, kw in enumerate((dict(reference=ref), dict(reference=ref), dict(), dict())):
    tm = {}
    ad = cnv.SimpleAnnData(X, var=var)
    t0 = time.perf_counter()
    cnv.tl.infercnv(ad, devices=[0], _timings=tm, **kw)
    dt = time.perf_counter() - t0
    print(f"cells {n}  {'reference given' if kw else 'reference=None  '}  {dt:7.3f} s  {n / dt / 1e6:6.3f} M cells/s  nnz {ad.obsm['X_cnv'].nnz}",
          {k: (round(x, 3) if isinstance(x, float) else x) for k, x in tm.items() if k in ("h2d", "reference_pass", "stream_and_kernels", "pack_pack_s", "csr_pack_d2h_tail", "sparse_upload")}, flush=True)
    del ad

#!/bin/bash
set -u
O=$PWD/gpurun_out/r05f; mkdir -p $O
for pb in 2e9 1e9 5e8; do
  ICV_PIECE_BYTES=$pb timeout 600 python - > $O/e2e_piece_$pb.txt 2>&1 <<PY
import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import numpy as np, pandas as pd, torch
import bench, cases
import infercnvpy_amd as cnv
X = bench.synth_rows(torch, 0, 200000, 20000).cpu().numpy()
torch.cuda.empty_cache()
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
ref = np.asarray(X[:2000].mean(axis=0), dtype=np.float64).astype(np.float32)
for rep in range(5):
    tm = {}
    t0 = time.perf_counter()
    kw = dict(reference=ref) if rep < 3 else dict()
    cnv.tl.infercnv(cnv.SimpleAnnData(X, var=var), devices=[0], _timings=tm, **kw)
    dt = time.perf_counter() - t0
    print("$pb", rep, round(dt, 4), round(200000 / dt), {k: (round(x, 4) if isinstance(x, float) else x) for k, x in tm.items() if k.startswith("pack_") or k in ("h2d", "stream_and_kernels", "reference_pass", "csr_pack_d2h_tail")})
PY
  tail -5 $O/e2e_piece_$pb.txt
done

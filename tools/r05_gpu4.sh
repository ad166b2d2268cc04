#!/bin/bash
# round 5, fourth GPU call: full GPU suite on the current tree, sparse upload timing (pool, thread counts), density lines of k_smooth_se
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/pytest_gpu.txt
cat /sys/kernel/mm/transparent_hugepage/enabled > $O/thp.txt 2>&1
for thr in 32 64 128; do
  ICV_PACK_THREADS=$thr timeout 600 python - > $O/e2e_pack_$thr.txt 2>&1 <<PY
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
for rep in range(3):
    tm = {}
    t0 = time.perf_counter()
    cnv.tl.infercnv(cnv.SimpleAnnData(X, var=var), reference=ref, devices=[0], _timings=tm)
    dt = time.perf_counter() - t0
    print(rep, round(dt, 4), round(200000 / dt), {k: (round(x, 4) if isinstance(x, float) else x) for k, x in tm.items() if k.startswith("pack_") or k in ("h2d", "stream_and_kernels", "sparse_upload")})
PY
  tail -3 $O/e2e_pack_$thr.txt
done
BASE="--no-cpu-baseline --no-e2e --no-extra"
for d in 0.02 0.07 0.14; do
  timeout 300 python bench.py --format csr --cells 500000 --window 250 --density $d --steps 10 --warmup 2 $BASE 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('density $d', 'ms/step', round(d['ms_per_step'],3), 'k_smooth_se ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), 'stages', {k:round(v,3) for k,v in d['stages']['kernel_ms'].items()})" | tee -a $O/csr_density_lines.txt
done
echo done

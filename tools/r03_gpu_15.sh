#!/bin/bash
# Round-3 GPU call: A/B of k_smooth_se builds: the package library ("new") against tools/variants/libinfercnv_hip_$N.so
# for every argument N (default: sebase = an earlier commit): CSR parity tests of the new build, then alternating
# timings (config 4; CSR window 100).
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03p; mkdir -p $O; rm -f $O/se_ab.txt
export TMPDIR=/tmp
VARIANTS="${*:-sebase}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or csr or sweep or order" 2>&1 | tail -2 | tee $O/pytest_new.txt
for rep in 1 2 3; do
  for v in $VARIANTS new; do
    if [ $v = new ]; then unset INFERCNV_HIP_LIB; else export INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_$v.so; fi
    for cfg in "250 500000" "100 200000"; do
      set -- $cfg
      timeout 200 python bench.py --format csr --cells $2 --window $1 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
      python -c "import json; d=json.load(open('$O/b.json')); print('$v rep $rep window $1: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],4))" | tee -a $O/se_ab.txt
    done
  done
done

#!/bin/bash
# Round-3 GPU call: static wave priority by age in k_smooth_se (0 none, 1 younger half, 2 graded, 3 last wavefront).
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    export INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_prio$v.so
    for cfg in "250 500000" "100 200000"; do
      set -- $cfg
      timeout 200 python bench.py --format csr --cells $2 --window $1 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
      python -c "import json; d=json.load(open('$O/b.json')); print('prio $v window $1 rep $rep: kernel', round(d['roofline']['kernel_ms'],4))" | tee -a $O/prio.txt
    done
  done
done
for t in 64 448; do
  echo "== prio 2, thread $t window 250" | tee -a $O/phase.txt
  ICV_PHASE_PROFILE=1 INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_prio2_prof$t.so timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "icv se profile" | tail -1 | tee -a $O/phase.txt
done

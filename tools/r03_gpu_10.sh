#!/bin/bash
# Round-3 GPU call: upper bounds for k_smooth_x16 (wrong-result experiment builds): what would a conflict-free scatter,
# no HBM row traffic, L2-resident rows, no x_res stores buy?  Alternating with the shipped library.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in shipped linscat noload l2row nostore noload_linscat; do
    if [ $v = shipped ]; then unset INFERCNV_HIP_LIB; else export INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_$v.so; fi
    timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-e2e --no-extra 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; d=json.load(open('$O/b.json')); print('$v rep $rep: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/x16_bounds.txt
  done
done

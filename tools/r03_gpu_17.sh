#!/bin/bash
# Round-3 GPU call: does a THIRD workgroup per CU pay?  Experiment builds of k_smooth_se for small geometries (<= 2048
# blocks, <= 1024 windows, rows of <= 1024 entries: four blocks per thread, two window slots, 43 KB of LDS), built from
# a scratch copy of the sources (not part of the tree): "small3" with 80 VGPRs (three workgroups per CU fit), "small2"
# the same source at 128 VGPRs.  Workload: CSR 500 000 x 20 000 at 3 % density, window 100 / step 20 (901 windows).
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03s; mkdir -p $O; rm -f $O/se_third_workgroup.txt
export TMPDIR=/tmp
run() {  # label, lib ("" = shipped), workgroups per CU ("" = default)
  if [ -n "$2" ]; then export INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_$2.so; else unset INFERCNV_HIP_LIB; fi
  if [ -n "$3" ]; then export ICV_WGS_PER_CU=$3; else unset ICV_WGS_PER_CU; fi
  timeout 200 python bench.py --format csr --cells 500000 --window 100 --step 20 --density 0.03 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('$1: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'][:40])" | tee -a $O/se_third_workgroup.txt
}
for rep in 1 2 3; do
  run "shipped (8 blocks/thread, 122 VGPRs), 2 per CU" "" ""
  run "small2 (128 VGPRs), 2 per CU" small2 ""
  run "small3 (80 VGPRs), 2 per CU" small3 2
  run "small3 (80 VGPRs), 3 per CU" small3 ""
  run "small3 (80 VGPRs), 1 per CU" small3 1
done

#!/bin/bash
# Round-3 GPU call: the default CSR geometry (window 100 / step 10 at 20 000 genes: 2000 blocks, 1802 windows) on an
# experiment build of k_smooth_se with four blocks per thread (43 KB of LDS) and 80 VGPRs (72-120 bytes of scratch):
# THREE workgroups per CU.  Built from a scratch copy of the sources (not part of the tree).
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03s; mkdir -p $O; rm -f $O/se_w100_three_workgroups.txt
export TMPDIR=/tmp
run() {  # label, lib ("" = shipped), workgroups per CU ("" = default)
  if [ -n "$2" ]; then export INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_$2.so; else unset INFERCNV_HIP_LIB; fi
  if [ -n "$3" ]; then export ICV_WGS_PER_CU=$3; else unset ICV_WGS_PER_CU; fi
  timeout 200 python bench.py --format csr --cells 200000 --window 100 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('$1: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],4))" | tee -a $O/se_w100_three_workgroups.txt
}
INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_w100x3.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or csr or sweep or order" 2>&1 | tail -2 | tee $O/pytest_w100x3.txt
for rep in 1 2 3; do
  run "shipped (122 VGPRs, 76 KB LDS), 2 per CU" "" ""
  run "experiment (80 VGPRs, 43 KB LDS), 2 per CU" w100x3 2
  run "experiment (80 VGPRs, 43 KB LDS), 3 per CU" w100x3 ""
done

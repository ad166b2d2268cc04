#!/bin/bash
# Round-3 GPU call: k_smooth_x16 with running row offsets + packed subtraction (inline asm) against the previous build.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or identical or benchmark_geometry or full_size or sweep" 2>&1 | tail -2 | tee $O/pytest.txt
for rep in 1 2 3 4; do
  for v in new base; do
    if [ $v = new ]; then unset INFERCNV_HIP_LIB; else export INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_x16base.so; fi
    timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-e2e --no-extra 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; d=json.load(open('$O/b.json')); print('$v rep $rep: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/x16_ab.txt
  done
done

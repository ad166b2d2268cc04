#!/bin/bash
# Round-3 GPU call: se kernel with the LDS-broadcast rank step; x16 with packed centring.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or sweep or benchmark_geometry or order or identical or full_size or ties" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
  for cfg in "csr 250 500000" "csr 100 200000" "dense 100 100000" "dense 250 100000"; do
    set -- $cfg
    timeout 200 python bench.py --format $1 --cells $3 --window $2 --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-extra 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; d=json.load(open('$O/b.json')); print('$1 window $2 cells $3 rep $rep: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/times.txt
  done
done
for t in 0 64 448; do
  for cfg in "250 500000"; do
    set -- $cfg
    echo "== thread $t window $1" | tee -a $O/phase.txt
    ICV_PHASE_PROFILE=1 INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_prof$t.so timeout 200 python bench.py --format csr --cells $2 --window $1 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "icv se profile" | tail -1 | tee -a $O/phase.txt
  done
done

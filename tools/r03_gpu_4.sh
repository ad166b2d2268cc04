#!/bin/bash
# Round-3 GPU call: re-pipelined k_smooth_se -- parity tests, timing against k_smooth_sd, phase timers.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
for rep in 1 2; do
  for cfg in "250 500000 0.07" "100 200000 0.07" "250 500000 0.02" "250 500000 0.14"; do
    set -- $cfg
    timeout 200 python bench.py --format csr --cells $2 --window $1 --density $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; d=json.load(open('$O/b.json')); print('se window $1 cells $2 density $3 rep $rep: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/se_times.txt
  done
done
for t in 0 64 448; do
  for cfg in "250 500000" "100 200000"; do
    set -- $cfg
    echo "== thread $t window $1" | tee -a $O/phase.txt
    ICV_PHASE_PROFILE=1 INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_prof$t.so timeout 200 python bench.py --format csr --cells $2 --window $1 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "icv se profile" | tail -1 | tee -a $O/phase.txt
  done
done

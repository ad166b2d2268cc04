#!/bin/bash
# Round-3 GPU call: k_smooth_se at one and two workgroups per CU (how the time scales with the wavefronts in flight).
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03q; mkdir -p $O; rm -f $O/se_occupancy.txt
export TMPDIR=/tmp
for rep in 1 2; do
  for n in 2 1; do
    for cfg in "250 500000" "100 200000"; do
      set -- $cfg
      ICV_WGS_PER_CU=$n timeout 200 python bench.py --format csr --cells $2 --window $1 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
      python -c "import json; d=json.load(open('$O/b.json')); print('workgroups per CU $n rep $rep window $1: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],4))" | tee -a $O/se_occupancy.txt
    done
  done
done

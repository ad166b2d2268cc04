#!/bin/bash
# Round-3 GPU call: k_smooth_se (second-generation stored-entries kernel) -- parity tests, then A/B against k_smooth_sd
# (ICV_SE_OFF=1) on config 4 (CSR window 250, 500 000 cells) and CSR window 100 (200 000 cells), two densities.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest.txt
for rep in 1 2; do
for v in se sd; do
  if [ $v = sd ]; then export ICV_SE_OFF=1; else unset ICV_SE_OFF; fi
  for cfg in "250 500000 0.07" "100 200000 0.07" "250 500000 0.02" "250 500000 0.14"; do
    set -- $cfg
    timeout 200 python bench.py --format csr --cells $2 --window $1 --density $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b_$v.json
    python -c "import json; d=json.load(open('$O/b_$v.json')); print('$v window $1 cells $2 density $3 rep $rep: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/se_vs_sd.txt
  done
done
done
unset ICV_SE_OFF
du -sh $O

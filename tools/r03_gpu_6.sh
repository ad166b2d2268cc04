#!/bin/bash
# Round-3 GPU call: se kernel with the wavefront-total reads skipped; column-sum access patterns; CSR upload experiments.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or sweep or benchmark_geometry or order or identical" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
  for cfg in "250 500000 0.07" "100 200000 0.07"; do
    set -- $cfg
    timeout 200 python bench.py --format csr --cells $2 --window $1 --density $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; d=json.load(open('$O/b.json')); print('se window $1 cells $2 density $3 rep $rep: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/se_times.txt
  done
done
timeout 300 tools/microbench_colsum.bin 2>&1 | tee $O/colsum_patterns.txt | tail -40
timeout 600 python tools/exp_h2d_csr.py 2>&1 | tee $O/h2d_csr.txt | tail -20

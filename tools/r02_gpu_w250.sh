#!/bin/bash
O=gpurun_out/${1:-r02w}; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
B="--steps 30 --warmup 5 --no-cpu-baseline --no-e2e"
timeout 120 python bench.py $B 2>&1 | tail -1 | one dense_w100 | tee -a $O/bench.txt
timeout 120 python bench.py $B --window 250 2>&1 | tail -1 | one dense_w250_x16 | tee -a $O/bench.txt
ICV_NO_X16=1 timeout 120 python bench.py $B --window 250 2>&1 | tail -1 | one dense_w250_ws | tee -a $O/bench.txt
timeout 200 python bench.py $B --format csr --cells 500000 --window 250 2>&1 | tail -1 | one csr_w250_500k | tee -a $O/bench.txt

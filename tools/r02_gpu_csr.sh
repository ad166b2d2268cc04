#!/bin/bash
# config 4 kernel: full GPU tests, then the CSR bench lines (window 250 at 500 000 cells, window 100 at 200 000)
O=gpurun_out/${1:-r02csr}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest.txt
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
for i in 1 2; do timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | one csr_w250 | tee -a $O/bench.txt; done
timeout 200 python bench.py --format csr --cells 200000 --window 100 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | one csr_w100 | tee -a $O/bench.txt

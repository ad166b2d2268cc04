#!/bin/bash
# pipelined k_colsum_csr: full GPU tests, then the config-4 step (reference means + smoothing + thresholds)
O=gpurun_out/${1:-r02colsum}; mkdir -p $O
timeout 800 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'cells/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'kernel_ms', round(r['kernel_ms'],3))"; }
C="--format csr --cells 500000 --window 250 --warmup 2 --no-cpu-baseline --no-e2e"
timeout 200 python bench.py $C --steps 5 2>$O/err.txt | tail -1 | one csr_w250 | tee $O/lines.txt
tail -3 $O/err.txt

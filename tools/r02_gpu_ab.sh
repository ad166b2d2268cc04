#!/bin/bash
# A/B of library variants on the default bench, alternating runs: tools/r02_gpu_ab.sh <tag> <variant.so> [n]
O=gpurun_out/${1:-r02ab}; mkdir -p $O
V=$2; N=${3:-3}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
for i in $(seq $N); do
  timeout 120 $B 2>/dev/null | tail -1 | one default | tee -a $O/ab.txt
  INFERCNV_HIP_LIB=$PWD/$V timeout 120 $B 2>/dev/null | tail -1 | one $V | tee -a $O/ab.txt
done

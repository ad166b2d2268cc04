#!/bin/bash
# same-box A/B of library variants built side by side (tools/libicv_*.so) on the config-4 bench line;
# the in-tree build is "base".  Two alternating rounds: boxes differ by +-3 %, only same-box numbers compare.
O=gpurun_out/${1:-r02ab}; mkdir -p $O
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4))"; }
C="--format csr --cells 500000 --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e"
for round in 1 2; do
  timeout 200 python bench.py $C 2>/dev/null | tail -1 | one base | tee -a $O/ab.txt
  for v in tools/libicv_*.so; do
    INFERCNV_HIP_LIB=$PWD/$v timeout 200 python bench.py $C 2>/dev/null | tail -1 | one $(basename $v .so) | tee -a $O/ab.txt
  done
done

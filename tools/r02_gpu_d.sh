#!/bin/bash
# quick A/B of build variants (timing only for the *_exp builds): every build twice, interleaved
set -u
O=gpurun_out/${1:-r02x}; mkdir -p $O
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
for rep in 1 2 3 4 5; do
  timeout 120 $B 2>&1 | tail -1 | one default | tee -a $O/bench.txt
  for lib in variants/libicv_*.so; do
    case $lib in *prof*) continue;; esac
    INFERCNV_HIP_LIB=$PWD/$lib timeout 120 $B 2>&1 | tail -1 | one $lib | tee -a $O/bench.txt
  done
done
if [ -f variants/libicv_prof.so ]; then INFERCNV_HIP_LIB=$PWD/variants/libicv_prof.so ICV_PHASE_PROFILE=1 timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "x16 profile" | tee $O/phase.txt; fi
exit 0

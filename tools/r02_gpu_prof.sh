#!/bin/bash
O=gpurun_out/${1:-r02x}; mkdir -p $O
for lib in variants/libicv_prof*.so; do
  for i in 1; do
  echo $lib | tee -a $O/phase.txt
  INFERCNV_HIP_LIB=$PWD/$lib ICV_PHASE_PROFILE=1 timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "x16 profile" | tee -a $O/phase.txt
  done
done
exit 0

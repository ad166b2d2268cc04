#!/bin/bash
# phase profile of k_smooth_sd (-DICV_SD_PROFILE build in tools/libicv_sdprof.so)
O=gpurun_out/${1:-r02sdp}; mkdir -p $O
INFERCNV_HIP_LIB=$PWD/tools/libicv_sdprof.so ICV_PHASE_PROFILE=1 timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e 2>&1 | grep "sd profile" | tail -2 | tee $O/phases.txt

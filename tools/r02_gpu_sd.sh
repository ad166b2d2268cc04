#!/bin/bash
# k_smooth_sd (CSR, long windows): the kernel-equivalence tests, then config 4 with and without it (ICV_NO_SD=1:
# the dense-row CSR kernel) on the same box, then the phase profile of a -DICV_SD_PROFILE build if one is present
O=gpurun_out/${1:-r02sd}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "identical or golden or sparse_input or csr" 2>&1 | tail -8 | tee $O/pytest.txt
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4))"; }
for i in 1 2; do
  timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>$O/err_sd.txt | tail -1 | one sd_w250 | tee -a $O/bench.txt
  ICV_NO_SD=1 timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | one ws_w250 | tee -a $O/bench.txt
done
tail -3 $O/err_sd.txt
if [ -f tools/libicv_sdprof.so ]; then
  INFERCNV_HIP_LIB=$PWD/tools/libicv_sdprof.so ICV_PHASE_PROFILE=1 timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e 2>&1 | grep "sd profile" | tail -2 | tee $O/phases.txt
fi

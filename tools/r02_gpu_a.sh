#!/bin/bash
# round 2, GPU call A: parity of the x16 kernel, A/B against k_smooth_ws on the same box, build variants, op costs
set -u
O=gpurun_out/r02a; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
timeout 120 tools/microbench_ops.bin > $O/ops.txt 2>&1; cat $O/ops.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
for i in 1 2; do timeout 120 $B 2>&1 | tail -1 | one x16_default | tee -a $O/bench.txt; done
ICV_NO_X16=1 timeout 120 $B 2>&1 | tail -1 | one ws_prev | tee -a $O/bench.txt
for lib in variants/libicv_wfirst2.so variants/libicv_wch2.so variants/libicv_wch10.so; do
  INFERCNV_HIP_LIB=$PWD/$lib timeout 120 $B 2>&1 | tail -1 | one $lib | tee -a $O/bench.txt
done
INFERCNV_HIP_LIB=$PWD/variants/libicv_prof.so ICV_PHASE_PROFILE=1 timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "x16 profile" | tee $O/phase.txt
timeout 200 python bench.py --cells 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | one x16_1M | tee -a $O/bench.txt

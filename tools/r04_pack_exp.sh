#!/bin/bash
# k_thr_pack upper-bound experiments (wrong-result builds): no look-back / no stores / neither, against the shipped library
O=gpurun_out/r04; mkdir -p $O
for v in "" nolb nost nolbst ""; do
  if [ -n "$v" ]; then export INFERCNV_HIP_LIB=$PWD/tools/variants/libinfercnv_hip_$v.so; else unset INFERCNV_HIP_LIB; fi
  timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${v:-shipped}', round(d['ms_per_step'],3), {k: round(x,3) for k,x in d['stages']['kernel_ms'].items()})"
done 2>&1 | tee $O/pack_experiments.txt

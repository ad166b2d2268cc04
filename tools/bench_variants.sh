#!/bin/bash
# Bench the default library and every variants/libicv_*.so (developer builds with other -D knobs) in one GPU call.
#   tools/bench_variants.sh [repeats]
REP=${1:-2}
for lib in "" variants/libicv_*.so; do
  for i in $(seq $REP); do
    INFERCNV_HIP_LIB=${lib:+$PWD/$lib} timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-default}', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done

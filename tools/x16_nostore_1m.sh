#!/bin/bash
# VERDICT r5 #4b: what the x_res stores cost k_smooth_x16 at 1 M cells (sustained clocks): the shipped library against
# the experiment build without the stores (tools/build_variant.sh x16_nostore -DICV_DEV_EXPERIMENTS -DICV_X_EXP_NOSTORE;
# wrong results on purpose), alternating, same box.
for i in 1 2; do
  for lib in "" tools/variants/libinfercnv_hip_x16_nostore.so; do
    INFERCNV_HIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --extra config3_cells_on_one_gpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['extra']['config3_cells_on_one_gpu']
print('${lib:-shipped}'.split('/')[-1], '100 000 cells: kernel', round(d['roofline']['kernel_ms'],3), 'ms frac', round(d['roofline']['frac'],3), '| 1 M cells: kernel', round(o['roofline']['kernel_ms'],3), 'ms frac', round(o['roofline']['frac'],3), 'step', round(o['ms_per_step'],2), 'ms') if 'roofline' in o else print('${lib:-shipped}', o)"
  done
done

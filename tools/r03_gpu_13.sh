#!/bin/bash
# Round-3 GPU call: k_thr_mask / k_csr_fill_masked restructured -- public-API parity, product-path leg, e2e legs.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --extra product_path_csr_pack 2>/dev/null | tail -1 > $O/b.json
  python - <<PY | tee -a $O/product.txt
import json
d=json.load(open('$O/b.json'))
p=d['extra']['product_path_csr_pack']
print('rep $rep product path', round(p['ms_per_step'],3), {k:round(v,4) for k,v in p['kernel_ms'].items()}, 'mask frac', round(p['roofline_k_thr_mask']['frac'],3), 'fill frac', round(p['roofline_k_csr_fill_masked']['frac'],3))
for k,e in d['e2e'].items(): print('   ', k[:40], round(e['seconds'],4), 's', round(e['cells_per_s']/1e6,3), 'M cells/s h2d', round(e['h2d_GBps'],1), 'drain busy', e['stages_s'].get('csr_pack_d2h'))
PY
done

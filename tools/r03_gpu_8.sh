#!/bin/bash
# Round-3 GPU call: two uploader threads per CSR slab -- public-API tests, e2e legs.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -x -k "golden or sweep or multi_slab or public_api or edge or error" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 20 2>/dev/null | tail -1 > $O/b.json
  python - <<PY | tee -a $O/e2e.txt
import json
d=json.load(open('$O/b.json'))
for k,e in d['e2e'].items(): print('rep $rep', k[:40], round(e['seconds'],4), 's', round(e['cells_per_s']/1e6,3), 'M cells/s h2d', round(e['h2d_GBps'],1), e['stages_s'])
PY
done

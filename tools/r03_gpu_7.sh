#!/bin/bash
# Round-3 GPU call: non-temporal column sums (1024-row slabs) in the bench step; prefaulted X_cnv arrays in the e2e legs.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or sweep or column_sums or multi_slab or full_size" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $O/b_pf.json
  ICV_NO_PREFAULT=1 timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $O/b_nopf.json
  for v in pf nopf; do
    python - <<PY | tee -a $O/e2e_ab.txt
import json
d=json.load(open('$O/b_$v.json'))
print('$v rep $rep step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3))
for k,e in d['e2e'].items(): print('   ', k[:40], round(e['seconds'],4), 's', round(e['cells_per_s']/1e6,3), 'M cells/s h2d', round(e['h2d_GBps'],1), e['stages_s'])
PY
  done
done

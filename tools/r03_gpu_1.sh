#!/bin/bash
# Round-3 GPU call 1: full GPU test-suite, the default bench line (with e2e + extra legs), the one-GPU dry run of the
# N-rank bench, CSR window-100 A/B (stored-entries kernel vs the row-in-LDS kernel).  Output: gpurun_out/r03a/.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $O/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
( time timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time; tail -c 400 $O/bench_n1.json; tail -3 $O/bench_n1.time
timeout 300 python bench.py --gpus 2 --dry-run-one-gpu --steps 3 --warmup 1 > $O/bench_dry2.json 2> $O/bench_dry2.err; tail -c 600 $O/bench_dry2.json; tail -5 $O/bench_dry2.err
for v in new old; do
  if [ $v = old ]; then export ICV_SD_MIN_NBW=10; else unset ICV_SD_MIN_NBW; fi
  timeout 200 python bench.py --format csr --cells 200000 --window 100 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/bench_csr_w100_$v.json
  python -c "import json; d=json.load(open('$O/bench_csr_w100_$v.json')); print('csr_w100 $v', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))" | tee -a $O/csr_w100_ab.txt
done
unset ICV_SD_MIN_NBW
du -sh $O

#!/bin/bash
# Round-3 GPU call: k_smooth_se with biased bins -- parity tests, timing, one phase profile, instruction counters.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
for rep in 1 2; do
  for cfg in "250 500000 0.07" "100 200000 0.07" "250 500000 0.02"; do
    set -- $cfg
    timeout 200 python bench.py --format csr --cells $2 --window $1 --density $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; d=json.load(open('$O/b.json')); print('se window $1 cells $2 density $3 rep $rep: step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))" | tee -a $O/se_times.txt
  done
done
for t in 64; do
  for cfg in "250 500000" "100 200000"; do
    set -- $cfg
    echo "== thread $t window $1" | tee -a $O/phase.txt
    ICV_PHASE_PROFILE=1 INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_prof$t.so timeout 200 python bench.py --format csr --cells $2 --window $1 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "icv se profile" | tail -1 | tee -a $O/phase.txt
  done
done
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$name -o pmc -- python $REPO/bench.py --format csr --cells 500000 --window 250 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_$name.log 2>&1)
  f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (echo "== --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f" | grep -A9 "k_smooth_se") | tee -a $O/pmc_summary.txt
done
rm -rf $O/pmc_SQ_INSTS_VALU

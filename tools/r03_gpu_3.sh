#!/bin/bash
# Round-3 GPU call: phase timers (three profiled wavefronts) and instruction counters of k_smooth_se.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
for t in 0 64 448; do
  for cfg in "250 500000" "100 200000"; do
    set -- $cfg
    echo "== thread $t window $1" | tee -a $O/phase.txt
    ICV_PHASE_PROFILE=1 INFERCNV_HIP_LIB=$REPO/tools/variants/libinfercnv_hip_prof$t.so timeout 200 python bench.py --format csr --cells $2 --window $1 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "icv se profile" | tail -1 | tee -a $O/phase.txt
  done
done
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
  name=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$name -o pmc -- python $REPO/bench.py --format csr --cells 500000 --window 250 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_$name.log 2>&1)
  f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (echo "== --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f" | grep -A9 "k_smooth_se") | tee -a $O/pmc_summary.txt
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete 2>/dev/null
rm -rf $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_ACTIVE_INST_VALU $O/pmc_SQ_INST_CYCLES_VMEM

#!/bin/bash
# Round-3: instruction-mix counters of the CSR kernels (config 4: window 250; CSR window 100), separate --pmc passes.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
for cfg in "250 500000" "100 200000"; do
  set -- $cfg; W=$1; C=$2
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $grp | cut -d' ' -f1)
    (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_${W}_$name -o pmc -- python $REPO/bench.py --format csr --cells $C --window $W --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_${W}_$name.log 2>&1)
    f=$(find $O/pmc_${W}_$name -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (echo "== window $W cells $C --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") | tee -a $O/pmc_summary.txt
  done
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete 2>/dev/null
rm -rf $O/pmc_*_SQ_INSTS_VALU $O/pmc_*_SQ_ACTIVE_INST_VALU $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
du -sh $O

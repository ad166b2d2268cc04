#!/bin/bash
# Extra PMC passes for the smoothing kernel (run on the GPU box): LDS / VMEM latency levels and FIFO stalls.
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmc_extra
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/summarize_pmc.py "$f" | grep -A8 "k_smooth_ws" | tee -a "$OUT/summary.txt"
done
find "$OUT" -name "*.csv" -size +1M -delete

#!/bin/bash
# Run on the GPU box (via gpurun): phase breakdown, rocprofv3 kernel stats, optional PMC passes.
# Usage: tools/gpu_profile.sh <tag> [pmc]
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"

echo "== phase profile" | tee "$OUT/phase.txt"
ICV_PHASE_PROFILE=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -A8 "icv phase" | tee -a "$OUT/phase.txt"

echo "== rocprofv3 kernel stats"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH > "$OUT/rocprof_stats.log" 2>&1)
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
head -12 "$OUT/kernel_stats.csv"

if [ "${2:-}" = "pmc" ]; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | cut -d' ' -f1)
    (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1)
    f=$(find "$OUT/pmc_$name" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $REPO/tools/summarize_pmc.py "$f" | tee "$OUT/pmc_$name.txt"
  done
fi
# keep only small text artefacts (gpurun_out merge limit)
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"

#!/bin/bash
# round 5, third GPU call: chain v3 (per-lane trash), sparse upload of dense host input (tests + e2e A/B)
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket|Flags" | cut -c1-400 > $O/box.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_refmean.py tests/test_gpu_resident_chain.py tests/test_gpu_sparse_upload.py -q 2>&1 | tail -15 | tee $O/pytest_a.txt
BASE="--no-cpu-baseline --no-e2e --no-extra"
C4="--format csr --cells 500000 --window 250"
timeout 300 python bench.py $C4 --steps 20 --warmup 3 $BASE > $O/bench_csr_w250_new.json 2> $O/bench_csr_w250_new.err
stats() {
  name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_s_$name -o bench -- python $REPO/bench.py "$@" > $O/bench_${name}_traced.json 2> $O/rocprof_$name.log)
  find $O/_s_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$name.csv
  rm -rf $O/_s_$name
  grep "icv::" $O/kernel_stats_$name.csv | head -6 | cut -c1-150
}
stats csr_w250 $C4 --steps 5 --warmup 2 $BASE
pmc() {
  label=$1; shift
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"; do
    name=$(echo $grp | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/_p -o pmc -- python $REPO/bench.py "$@" > $O/pmc_${label}_$name.log 2>&1)
    f=$(find $O/_p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (echo "== $label --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") >> $O/pmc_summary.txt
    rm -rf $O/_p
  done
}
pmc csr_w250_500000_cells $C4 --steps 2 --warmup 1 $BASE
grep -A8 "colchain_csrq" $O/pmc_summary.txt | head -40
timeout 600 python bench.py --steps 50 --no-cpu-baseline --no-extra > $O/bench_e2e_sparse.json 2> $O/bench_e2e_sparse.err
ICV_NO_SPARSE_UPLOAD=1 timeout 600 python bench.py --steps 50 --no-cpu-baseline --no-extra > $O/bench_e2e_dense.json 2> $O/bench_e2e_dense.err
python - <<'PY'
import json
for n in ("sparse","dense"):
    d=json.loads(open(f"gpurun_out/r05c/bench_e2e_{n}.json").read().strip().splitlines()[-1])
    for k,v in d["e2e"].items(): print(n, k[:70], round(v["seconds"],4), round(v["cells_per_s"]), v["stages_s"])
PY
find $O -name "*.db" -delete 2>/dev/null
echo done

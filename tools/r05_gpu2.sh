#!/bin/bash
# round 5, second GPU call: v2 of the CSR chain kernels, F-ordered means, the fixed tests, untraced config-4 timing, PMC
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $O/box.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_refmean.py tests/test_gpu_resident_chain.py -x -q 2>&1 | tail -15 | tee $O/pytest_a.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "entry_order or golden_reference_mean or refmean" 2>&1 | tail -8 | tee $O/pytest_b.txt
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_pack.py tests/test_integration_stub.py -q 2>&1 | tail -8 | tee $O/pytest_c.txt
BASE="--no-cpu-baseline --no-e2e --no-extra"
C4="--format csr --cells 500000 --window 250"
timeout 300 python bench.py $C4 --steps 20 --warmup 3 $BASE > $O/bench_csr_w250_new.json 2> $O/bench_csr_w250_new.err
ICV_NO_CHAIN_QUEUES=1 timeout 300 python bench.py $C4 --steps 20 --warmup 3 $BASE > $O/bench_csr_w250_old.json 2> $O/bench_csr_w250_old.err
stats() {
  name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_s_$name -o bench -- python $REPO/bench.py "$@" > $O/bench_${name}_traced.json 2> $O/rocprof_$name.log)
  find $O/_s_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$name.csv
  rm -rf $O/_s_$name
  head -9 $O/kernel_stats_$name.csv | cut -c1-150
}
stats csr_w250 $C4 --steps 5 --warmup 2 $BASE
pmc() {
  label=$1; shift
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
    name=$(echo $grp | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/_p -o pmc -- python $REPO/bench.py "$@" > $O/pmc_${label}_$name.log 2>&1)
    f=$(find $O/_p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (echo "== $label --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") | tee -a $O/pmc_summary.txt | grep -i "colchain\|tile_bounds\|==" | head -12
    rm -rf $O/_p
  done
}
pmc csr_w250_500000_cells $C4 --steps 2 --warmup 1 $BASE
( time timeout 1200 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time; tail -c 400 $O/bench_n1.json; tail -3 $O/bench_n1.time
timeout 300 python bench.py --gpus 2 --dry-run-one-gpu --steps 3 --warmup 1 2>$O/dry2.err | tail -1 > $O/bench_dry_run_2ranks_one_gpu.json; tail -c 300 $O/bench_dry_run_2ranks_one_gpu.json
find $O -name "*.db" -delete 2>/dev/null
echo done

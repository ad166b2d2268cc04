#!/bin/bash
# Evidence run on the GPU box (one gpurun call): tests, smoke, the default bench line (stages, e2e, extra legs), rocprofv3
# kernel statistics of the bench step / config 4 / CSR window 100 / config 5, PMC passes (FETCH_SIZE and WRITE_SIZE in
# separate runs, --kernel-trace only; instruction mix), the N-rank dry run on one GPU.
#   ROUND=r06 tools/round_end.sh [quick]      -> gpurun_out/${ROUND}final/   (copy what is to be judged into profiles/)
set -u
REPO=$PWD
ROUND=${ROUND:-r06}
O=$REPO/gpurun_out/${ROUND}final; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $O/box.txt 2>&1
if [ "${1:-}" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
fi
( time timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time; tail -c 300 $O/bench_n1.json; tail -3 $O/bench_n1.time
timeout 300 python bench.py --gpus 2 --dry-run-one-gpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_dry_run_2ranks_one_gpu.json
timeout 600 python bench.py --gpus 8 --dry-run-one-gpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_dry_run_8ranks_one_gpu.json
stats() {  # name, bench.py arguments...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_s_$name -o bench -- python $REPO/bench.py "$@" > $O/bench_${name}_traced.json 2> $O/rocprof_$name.log)
  find $O/_s_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$name.csv
  rm -rf $O/_s_$name
  head -6 $O/kernel_stats_$name.csv | cut -c1-150
}
BASE="--no-cpu-baseline --no-e2e --no-extra"
stats dense_w100 --steps 20 --warmup 3 $BASE
stats csr_w250 --format csr --cells 500000 --window 250 --steps 5 --warmup 2 $BASE
stats csr_w100 --format csr --cells 200000 --window 100 --steps 5 --warmup 2 $BASE
stats config5 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --extra config5
stats gene_values_scores --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --extra scores_and_gene_values
pmc() {  # label, bench.py arguments...
  label=$1; shift
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/_p -o pmc -- python $REPO/bench.py "$@" > $O/pmc_${label}_$name.log 2>&1)
    f=$(find $O/_p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (echo "== $label --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") | tee -a $O/pmc_summary.txt | head -40
    rm -rf $O/_p
  done
}
pmc dense_w100_100000_cells --steps 2 --warmup 1 $BASE
pmc csr_w250_500000_cells --format csr --cells 500000 --window 250 --steps 2 --warmup 1 $BASE
pmc csr_w100_200000_cells --format csr --cells 200000 --window 100 --steps 2 --warmup 1 $BASE
pmc_hbm() {  # label, bench.py arguments...: HBM traffic only (two passes)
  label=$1; shift
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/_p -o pmc -- python $REPO/bench.py "$@" > $O/pmc_${label}_$grp.log 2>&1)
    f=$(find $O/_p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (echo "== $label --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") | tee -a $O/pmc_summary.txt | head -30
    rm -rf $O/_p
  done
}
pmc_hbm gene_values_100000_cells --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --extra scores_and_gene_values
# round 6: the CSR means over the density, the chain by integer blocks (one rank's share), the HBM write ceiling, clocks and
# power over the sustained 1 M-cell leg
timeout 400 python tools/time_csr_means.py 2>/dev/null | tee $O/csr_means_density.txt
timeout 300 python tools/time_chain_blocks.py 2>/dev/null | tee $O/chain_blocks_times.txt
timeout 120 python tools/time_hbm_write.py 2>/dev/null | tee $O/hbm_write.txt
timeout 200 python tools/time_gene_kernel.py 2>/dev/null | tail -1 | tee $O/gene_kernel.txt
BENCH_E2E_LEGS=1m timeout 400 python tools/host_issue_after_e2e.py 2>/dev/null | grep -v "amdgpu.ids" | tail -12 > $O/host_issue_after_e2e.txt; tail -8 $O/host_issue_after_e2e.txt | cut -c1-200
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)|Average Graphics" | tr -s " " | tr "\n" ";"; echo; sleep 0.25; done > $O/clocks_1m_leg.txt ) &
CLK=$!
timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-e2e --extra config3_cells_on_one_gpu > $O/bench_1m_leg.json 2> $O/bench_1m_leg.err
kill $CLK 2>/dev/null
tail -1 $O/bench_1m_leg.err | cut -c1-600
# round 5: k_smooth_se at three densities, the host-side packing of the sparse upload, fuzzers and the soak on the final tree
for d in 0.02 0.07 0.14; do
  timeout 300 python bench.py --format csr --cells 500000 --window 250 --density $d --steps 10 --warmup 2 $BASE 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('density $d', 'ms/step', round(d['ms_per_step'],3), 'k_smooth_se ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), 'stages', {k:round(v,3) for k,v in d['stages']['kernel_ms'].items()})" | tee -a $O/csr_density_lines.txt
done
timeout 300 python tools/bench_host_pack.py 50000 > $O/host_pack.txt 2>&1; tail -8 $O/host_pack.txt
if [ "${1:-}" != "quick" ]; then
  (echo "== tests/fuzz_gpu.py 11000 600"; timeout 900 python tests/fuzz_gpu.py 11000 600 2>&1 | tail -2
   echo "== tests/fuzz_gpu_big.py 600 120"; timeout 600 python tests/fuzz_gpu_big.py 600 120 2>&1 | tail -2
   echo "== tests/fuzz_gpu_ward.py 100 40"; timeout 300 python tests/fuzz_gpu_ward.py 100 40 2>&1 | tail -2
   echo "== tests/fuzz_gpu_means.py 7000 300"; timeout 900 python tests/fuzz_gpu_means.py 7000 300 2>&1 | tail -3
   echo "== tests/fuzz_gpu_means.py 100000 200 (dense float32: every case also through the chain by blocks)"; timeout 900 python tests/fuzz_gpu_means.py 100000 200 2>&1 | tail -3) | tee $O/fuzz_gpu.txt
  timeout 600 python tests/soak_gpu.py 200 2>&1 | tail -6 | tee $O/soak_public_api.txt
fi
find $O -name "*.db" -delete 2>/dev/null

#!/bin/bash
# Round-3 evidence run on the GPU box: tests, smoke, default bench line (e2e + extra legs), rocprofv3 kernel stats of the
# bench and of config 4, PMC passes (FETCH_SIZE / WRITE_SIZE separately, --kernel-trace only; instruction mix), the
# N-rank dry run on one GPU.  Everything lands in gpurun_out/r03final/.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r03final; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $O/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
( time timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time; tail -c 300 $O/bench_n1.json; tail -3 $O/bench_n1.time
timeout 300 python bench.py --gpus 2 --dry-run-one-gpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_dry_run_2ranks_one_gpu.json
B="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-extra"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_traced.json 2> $O/rocprof_stats.log)
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -8 $O/kernel_stats.csv
C4="python $REPO/bench.py --format csr --cells 500000 --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -o bench -- $C4 > $O/bench_csr_w250_traced.json 2> $O/rocprof_stats4.log)
find $O/stats4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_csr_w250.csv
head -8 $O/kernel_stats_csr_w250.csv
CW="python $REPO/bench.py --format csr --cells 200000 --window 100 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsw -o bench -- $CW > $O/bench_csr_w100_traced.json 2> $O/rocprof_statsw.log)
find $O/statsw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_csr_w100.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$name -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/pmc_$name.log 2>&1)
  f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (echo "== dense window 100, 100000 cells --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") | tee -a $O/pmc_summary.txt
  for cfg in "250 500000" "100 200000"; do
    set -- $cfg
    (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$1_$name -o pmc -- python $REPO/bench.py --format csr --cells $2 --window $1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_$1_$name.log 2>&1)
    f=$(find $O/pmc_$1_$name -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (echo "== CSR window $1, $2 cells --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") | tee -a $O/pmc_summary.txt
  done
done
timeout 200 python bench.py --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-extra 2>/dev/null | tail -1 > $O/bench_dense_w250.json
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete 2>/dev/null
rm -rf $O/stats $O/stats4 $O/statsw $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_ACTIVE_INST_VALU $O/pmc_250_* $O/pmc_100_*
rm -f $O/pmc_*.log
du -sh $O

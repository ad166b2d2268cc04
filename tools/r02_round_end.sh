#!/bin/bash
# Round-2 evidence run on the GPU box: tests, smoke, default bench line, rocprofv3 kernel stats of the bench,
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate --pmc runs, --kernel-trace only), the other BASELINE
# configurations, config 5 at full size with a kernel trace.  Everything lands in gpurun_out/r02final/.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r02final; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 | tee $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json
B="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_traced.json 2> $O/rocprof_stats.log)
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -8 $O/kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$name -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_$name.log 2>&1)
  f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (echo "== --pmc $grp"; python $REPO/tools/summarize_pmc.py "$f") | tee -a $O/pmc_summary.txt
done
timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/bench_csr_w250.json
timeout 200 python bench.py --window 250 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/bench_dense_w250.json
timeout 300 python bench.py --cells 1000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/bench_1m.json
for f in csr_w250 dense_w250 1m; do python -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernel'][:40], round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; done | tee $O/other_configs.txt
# config 5: full size un-traced, 100k traced (per-round durations)
timeout 600 python tools/bench_ward.py --cells 200000 2>&1 | tail -1 | tee $O/config5_200k.json
timeout 600 bash tools/profile_ward.sh 100000 > $O/ward_profile.log 2>&1
cp gpurun_out/ward_prof/rounds.txt $O/config5_ward_rounds_100k.txt 2>/dev/null
cp gpurun_out/ward_prof/ward_kernel_stats.csv $O/config5_kernel_stats_100k.csv 2>/dev/null
timeout 200 tools/bench_gram.bin 32768 5008 3 2>&1 | tee $O/gram_experiments.txt | tail -6
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete 2>/dev/null
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_ACTIVE_INST_VALU
du -sh $O

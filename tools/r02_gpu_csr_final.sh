#!/bin/bash
# Evidence run after the CSR long-window kernel (k_smooth_sd) went in: full GPU tests, the default bench line (with
# the e2e legs), config 4 at three densities (cost follows the stored entries), the short-window CSR kernel, and a
# rocprofv3 kernel trace of config 4.  Everything lands in gpurun_out/r02csr/.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r02csr; mkdir -p $O
export TMPDIR=/tmp
[ -n "${SKIP_TESTS:-}" ] || timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 900 $O/bench_n1.json
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'cells/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],4))"; }
C="--format csr --cells 500000 --window 250 --warmup 2 --no-cpu-baseline --no-e2e"
timeout 200 python bench.py $C --steps 5 2>/dev/null | tail -1 > $O/bench_csr_w250.json; one csr_w250_d0.07 < $O/bench_csr_w250.json | tee $O/csr_lines.txt
timeout 200 python bench.py $C --steps 3 --density 0.02 2>/dev/null | tail -1 | one csr_w250_d0.02 | tee -a $O/csr_lines.txt
timeout 200 python bench.py $C --steps 3 --density 0.14 2>/dev/null | tail -1 | one csr_w250_d0.14 | tee -a $O/csr_lines.txt
ICV_NO_SD=1 timeout 200 python bench.py $C --steps 3 2>/dev/null | tail -1 | one csr_w250_d0.07_row_in_lds | tee -a $O/csr_lines.txt
timeout 200 python bench.py --format csr --cells 200000 --window 100 --steps 3 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | one csr_w100_200k | tee -a $O/csr_lines.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o csr -- python $REPO/bench.py $C --steps 3 > $O/bench_csr_traced.json 2> $O/rocprof.log)
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_csr_w250.csv
grep -E "icv::" $O/kernel_stats_csr_w250.csv | cut -c1-170 | head -8
rm -rf $O/stats

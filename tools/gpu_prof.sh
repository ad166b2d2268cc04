#!/bin/bash
# rocprofv3 kernel statistics of one bench.py command on the GPU box:
#   tools/gpu_prof.sh OUT_NAME [bench.py arguments...]      -> gpurun_out/<round>/OUT_NAME_kernel_stats.csv (+ the bench line)
#   ROUND=r04 (default) names the output directory.
set -u
REPO=$PWD
O=$REPO/gpurun_out/${ROUND:-r04}; mkdir -p $O
name=$1; shift
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_stats_$name -o bench -- python $REPO/bench.py "$@" > $O/${name}_bench_traced.json 2> $O/${name}_rocprof.log)
find $O/_stats_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
rm -rf $O/_stats_$name
head -${LINES_SHOWN:-12} $O/${name}_kernel_stats.csv | cut -c1-160

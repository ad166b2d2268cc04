#!/bin/bash
# per-kernel durations of config 4 (rocprofv3 kernel trace): k_smooth_sd path, then the dense-row CSR kernel
O=$PWD/gpurun_out/${1:-r02sdprof}; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in sd ws; do
  if [ $v = ws ]; then export ICV_NO_SD=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o p -- python $R/bench.py --format csr --cells 500000 --window 250 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/$v.log 2>&1
  f=$(find $O/$v -name 'p_kernel_stats.csv' | head -1)
  echo "== $v"; head -8 $f | cut -c1-200
done

#!/bin/bash
# round 5, first GPU call: the new CSR chain kernel + the API closure tests, then the full suite, then config 4 A/B
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $O/box.txt 2>&1
timeout 420 python -m pytest tests/test_gpu_refmean.py -x -q 2>&1 | tail -15 | tee $O/pytest_refmean.txt
timeout 600 python -m pytest tests/test_gpu_resident_chain.py tests/test_gpu_resident.py -q 2>&1 | tail -40 | tee $O/pytest_resident.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_refmean.py --deselect tests/test_gpu_resident_chain.py --deselect tests/test_gpu_resident.py 2>&1 | tail -25 | tee $O/pytest_rest.txt
BASE="--no-cpu-baseline --no-e2e --no-extra"
stats() {  # name, env..., -- bench args
  name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_s_$name -o bench -- python $REPO/bench.py "$@" > $O/bench_${name}.json 2> $O/rocprof_$name.log)
  find $O/_s_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$name.csv
  rm -rf $O/_s_$name
  head -8 $O/kernel_stats_$name.csv | cut -c1-160
}
stats csr_w250_new --format csr --cells 500000 --window 250 --steps 5 --warmup 2 $BASE
ICV_NO_CHAIN_QUEUES=1 stats csr_w250_old --format csr --cells 500000 --window 250 --steps 5 --warmup 2 $BASE
stats dense_w100 --steps 20 --warmup 3 $BASE
find $O -name "*.db" -delete 2>/dev/null
echo done

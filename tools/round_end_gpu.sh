set -u
mkdir -p gpurun_out/final
timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 | tee gpurun_out/final/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; tail -c 1500 gpurun_out/final/bench_default.json
timeout 400 tools/gpu_profile.sh r01g pmc > gpurun_out/final/profile.log 2>&1; tail -5 gpurun_out/final/profile.log
timeout 200 python bench.py --format csr --cells 500000 --window 250 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final/bench_csr_w250.json
timeout 200 python bench.py --window 250 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final/bench_dense_w250.json
timeout 300 python bench.py --cells 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final/bench_1m.json
for f in csr_w250 dense_w250 1m; do python -c "import json,sys; d=json.load(open('gpurun_out/final/bench_$f.json')); print('$f', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; done

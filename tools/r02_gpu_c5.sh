#!/bin/bash
# config 5 on the GPU box: tests of the distance / Ward / ithcna path, then timings at 50k..200k cells
O=gpurun_out/${1:-r02c5}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "ward or linkage or pairwise or ith or corr or heatmap or sharded" 2>&1 | tail -12 | tee $O/pytest.txt
timeout 900 python tools/bench_ward.py --cells ${CELLS:-50000 100000 200000} 2>&1 | tee $O/ward.txt

#!/bin/bash
# GPU tests, then the variants A/B
O=gpurun_out/${1:-r02x}; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
bash tools/r02_gpu_d.sh $1

#!/bin/bash
O=gpurun_out/${1:-r02x}; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
bash tools/r02_gpu_d.sh $1
bash tools/r02_gpu_prof.sh $1

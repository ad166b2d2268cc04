#!/bin/bash
# Build a side-by-side variant of the library for A/B timing / profiling on the GPU box:
#   tools/build_variant.sh NAME [extra hipcc flags...]   ->  tools/variants/libinfercnv_hip_NAME.so
# (git-ignored, travels with gpurun; select it with INFERCNV_HIP_LIB=tools/variants/libinfercnv_hip_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 \
  -Wall -Wno-unused-function "$@" -o tools/variants/libinfercnv_hip_$name.so infercnvpy_amd/csrc/icv_api.hip
echo tools/variants/libinfercnv_hip_$name.so

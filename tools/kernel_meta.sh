#!/bin/bash
# Per-kernel resource metadata of the gfx950 code object (VGPRs, SGPRs, scratch bytes, LDS): device-only compile of
# icv_api.hip + llvm-readelf --notes.   tools/kernel_meta.sh [pattern]  -> name vgpr sgpr scratch lds
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/icv_meta
[ -n "${ICV_META_REUSE:-}" ] && [ -f /tmp/icv_meta/dev.o ] || \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 \
  --cuda-device-only -c -o /tmp/icv_meta/dev.o infercnvpy_amd/csrc/icv_api.hip ${ICV_EXTRA_HIPCC_FLAGS:-}
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/icv_meta/dev.o \
  --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/icv_meta/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/icv_meta/dev.co > /tmp/icv_meta/notes.txt
python3 - "$@" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
txt = open("/tmp/icv_meta/notes.txt").read()
for blk in txt.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    try:
        import subprocess
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    if pat and pat not in name:
        continue
    print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
PY

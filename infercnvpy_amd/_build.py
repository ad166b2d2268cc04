"""Build the gfx950 C-ABI shared library in-tree with hipcc (no cmake, no JIT cache).

    python -m infercnvpy_amd._build [--force]

Output: ``infercnvpy_amd/libinfercnv_hip.so`` (git-ignored; travels to the GPU box with the
repo snapshot).  hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libinfercnv_hip.so")
SOURCES = [os.path.join(CSRC, "icv_api.hip")]
DEPS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + [
    os.path.join(os.path.dirname(HERE), "include", "infercnv_hip.h"),
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-ffp-contract=off",  # float64 evaluation order is part of the parity contract
        "-mllvm", "-amdgpu-mfma-vgpr-form=1",  # k_gram_mfma keeps both accumulator levels in VGPRs
        "-Wall", "-Wno-unused-function",
        "-o", LIB + ".tmp",
    ] + os.environ.get("ICV_EXTRA_HIPCC_FLAGS", "").split() + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

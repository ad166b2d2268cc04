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


def _accepts(hipcc: str, flags: list[str]) -> bool:
    """Does this hipcc take `flags`?  (An -mllvm option the bundled LLVM does not know is a fatal error.)"""
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.hip")
        with open(src, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n__global__ void probe() {}\n")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-c", *flags, "-o", os.path.join(d, "probe.o"), src],
                           capture_output=True)
        return r.returncode == 0


def _extra_flags() -> list[str]:
    """ICV_EXTRA_HIPCC_FLAGS for tuning builds.  The upper-bound experiments of the kernels (code behind
    ICV_DEV_EXPERIMENTS: they produce WRONG results on purpose) never go into the package's library: they are built
    side by side with tools/build_variant.sh."""
    flags = os.environ.get("ICV_EXTRA_HIPCC_FLAGS", "").split()
    if any("ICV_DEV_EXPERIMENTS" in f or "_EXP_" in f for f in flags):
        raise RuntimeError("ICV_EXTRA_HIPCC_FLAGS: experiment switches (ICV_DEV_EXPERIMENTS / *_EXP_*) are refused for "
                           "the package library; use tools/build_variant.sh")
    return flags


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    # k_gram_mfma keeps both accumulator levels in VGPRs with this hint (0.80 -> 0.91 of the MFMA peak); it is only a
    # performance hint, so toolchains whose LLVM does not have the option build without it
    vgpr_form = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
    if not _accepts(hipcc, vgpr_form):
        vgpr_form = []
    cmd = [
        hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-ffp-contract=off",  # float64 evaluation order is part of the parity contract
        *vgpr_form,
        "-Wall", "-Wno-unused-function",
        "-o", LIB + ".tmp",
    ] + _extra_flags() + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

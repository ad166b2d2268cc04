"""ctypes binding of ``libinfercnv_hip.so`` (the C ABI in ``include/infercnv_hip.h``).

The product path has NO CPU fallback: if the shared library is missing, or a compute entry
point is called without a GPU, this module raises.  ``load()`` itself works on a GPU-less host
(hipcc cross-compiles; the HIP runtime loads without a device), which lets the planning entry
points -- pure host code -- be exercised by the CPU test-suite.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# INFERCNV_HIP_LIB: developer override (kernel variants built side by side for A/B timing)
LIB_PATH = os.environ.get("INFERCNV_HIP_LIB") or os.path.join(_HERE, "libinfercnv_hip.so")

ICV_OK, ICV_ERR_INVALID, ICV_ERR_UNSUPPORTED, ICV_ERR_HIP, ICV_ERR_NOMEM = range(5)
ICV_F32, ICV_F64 = 0, 1
ICV_DENSE, ICV_CSR = 0, 1
ICV_FLAG_TRUNC_TO_INT = 1
ICV_FLAG_ROUND_F32 = 2
ICV_FLAG_NO_APPLY = 4
(ICV_KERNEL_NONE, ICV_KERNEL_GENERIC, ICV_KERNEL_WS, ICV_KERNEL_WS_CSR, ICV_KERNEL_X16, ICV_KERNEL_SD,
 ICV_KERNEL_SPLIT) = range(7)

# every symbol include/infercnv_hip.h declares
EXPORTS = (
    "icv_plan_create", "icv_plan_destroy", "icv_plan_get_info", "icv_plan_chr_pos", "icv_plan_window_table",
    "icv_plan_last_kernel", "icv_plan_se_tables", "icv_plan_gene_runs",
    "icv_colsum", "icv_colchain", "icv_colsum_pairwise", "icv_colchain_mean", "icv_colmean_csc", "icv_infercnv_smooth", "icv_chunk_thresholds", "icv_apply_threshold", "icv_infercnv_run",
    "icv_infercnv_run_windows", "icv_gene_values_from_windows",
    "icv_colchain_blocks_workspace", "icv_colchain_blocks_sums", "icv_colchain_blocks_records", "icv_colchain_blocks_scan",
    "icv_profile_begin", "icv_profile_collect",
    "icv_gene_values", "icv_csr_count", "icv_csr_fill", "icv_threshold_mask", "icv_csr_fill_masked", "icv_row_offsets", "icv_pack_geometry", "icv_threshold_pack", "icv_corr_iqr",
    "icv_pairwise_sqeuclidean", "icv_ward_linkage", "icv_pairwise_sqeuclidean_tiles",
    "icv_ward_create", "icv_ward_destroy", "icv_ward_merge", "icv_ward_gather", "icv_ward_scatter", "icv_ward_scan", "icv_ward_pack_nn",
    "icv_ward_unpack_nn", "icv_ward_pairs", "icv_ward_round_pairs", "icv_ward_finish", "icv_row_abs_sum", "icv_csr_row_abs_sum", "icv_group_sums", "icv_csr_check", "icv_csr_densify", "icv_host_dense_row_nnz", "icv_host_dense_pack", "icv_host_dense_pack_fused", "icv_csr_scatter_dense", "icv_last_error", "icv_version",
    "icv_device_count", "icv_developer_knobs_reload",
)


class Matrix(C.Structure):
    _fields_ = [
        ("format", C.c_int32), ("dtype", C.c_int32), ("n_rows", C.c_int64), ("n_cols", C.c_int32),
        ("_pad", C.c_int32), ("ld", C.c_int64), ("values", C.c_void_p), ("indptr", C.c_void_p),
        ("indices", C.c_void_p), ("csr_begin", C.c_int64), ("csr_end", C.c_int64),
    ]


class PlanInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_cols_all", "n_genes_used", "n_chr", "window", "step", "n_windows", "block", "n_blocks", "padded_len",
        "lds_bytes_f32", "lds_bytes_f64", "workgroups_per_cu_f32")]


class PackInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "rows_per_round", "mask_streamed", "mask_ring_slots", "mask_lds_bytes", "mask_loads_in_flight", "fill_streamed",
        "fill_ring_slots", "fill_lds_bytes", "fill_loads_in_flight", "fill_stage_entries")]


class Profile(C.Structure):
    _fields_ = [("smooth_ms", C.c_float), ("thresholds_ms", C.c_float), ("apply_ms", C.c_float),
                ("total_ms", C.c_float)]


class HipExtensionMissing(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (raises HipExtensionMissing if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `python -m infercnvpy_amd._build` "
            "(hipcc --offload-arch=gfx950).  infercnvpy_amd has no CPU fallback."
        )
    # PyTorch ships its own copy of the HIP runtime (same SONAME as /opt/rocm's): it has to be the one the
    # process loads first, otherwise this library pulls in the system copy, torch is bound to it as well, and
    # device allocations from here fail with hipErrorNoDevice.  torch is a hard dependency (device memory,
    # streams) anyway.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    i32, i64, vp, dbl = C.c_int32, C.c_int64, C.c_void_p, C.c_double
    P = C.POINTER
    lib.icv_plan_create.argtypes = [i32, vp, i32, vp, i32, i32, P(vp)]
    lib.icv_plan_destroy.argtypes = [vp]
    lib.icv_plan_destroy.restype = None
    lib.icv_plan_get_info.argtypes = [vp, P(PlanInfo)]
    lib.icv_plan_chr_pos.argtypes = [vp, vp]
    lib.icv_plan_window_table.argtypes = [vp, vp, vp]
    lib.icv_plan_last_kernel.argtypes = [vp, P(i32)]
    lib.icv_plan_se_tables.argtypes = [vp, P(i32), vp, vp, vp, vp, vp]
    lib.icv_plan_gene_runs.argtypes = [vp, P(i32), vp, vp, vp, vp]
    lib.icv_colsum.argtypes = [P(Matrix), vp, i32, vp, vp]
    lib.icv_colchain.argtypes = [P(Matrix), vp, i64, dbl, vp, vp]
    lib.icv_colsum_pairwise.argtypes = [vp, i32, i64, i32, i64, vp, vp]
    lib.icv_colchain_mean.argtypes = [vp, i32, i32, i64, vp, vp]
    lib.icv_colmean_csc.argtypes = [vp, i32, vp, vp, i32, vp, i32, dbl, vp, vp]
    lib.icv_infercnv_smooth.argtypes = [vp, P(Matrix), vp, vp, dbl, i32, vp, i64, vp, vp, vp]
    lib.icv_chunk_thresholds.argtypes = [vp, i64, i64, i64, i32, dbl, vp, vp]
    lib.icv_apply_threshold.argtypes = [vp, P(Matrix), vp, vp, dbl, i32, vp, i64, vp, vp, i64, i64, vp]
    lib.icv_infercnv_run.argtypes = [vp, P(Matrix), vp, vp, dbl, dbl, i64, i64, i32, vp, i64, vp, vp, vp,
                                     P(Profile), vp]
    lib.icv_infercnv_run_windows.argtypes = [vp, P(Matrix), vp, vp, dbl, dbl, i64, i64, i32, vp, i64, vp, vp, vp,
                                             P(Profile), vp, i64, vp]
    lib.icv_gene_values_from_windows.argtypes = [vp, vp, i64, i64, vp, i64, i64, vp, i64, vp]
    lib.icv_colchain_blocks_workspace.argtypes = [i64, i32, P(i64)]
    lib.icv_colchain_blocks_sums.argtypes = [P(Matrix), vp, vp, vp]
    lib.icv_colchain_blocks_records.argtypes = [P(Matrix), vp, vp, vp]
    lib.icv_colchain_blocks_scan.argtypes = [P(Matrix), vp, vp, i32, i32, vp, vp]
    lib.icv_profile_begin.argtypes = [vp]
    lib.icv_profile_collect.argtypes = [vp, P(Profile), i32, P(i32)]
    lib.icv_gene_values.argtypes = [vp, P(Matrix), vp, vp, dbl, i32, vp, i64, i64, vp, i64, vp]
    lib.icv_csr_count.argtypes = [vp, i64, i32, i64, vp, vp]
    lib.icv_csr_fill.argtypes = [vp, i64, i32, i64, vp, vp, vp, vp]
    lib.icv_threshold_mask.argtypes = [vp, P(Matrix), vp, vp, dbl, i32, vp, i64, vp, vp, i64, i64, vp, vp, vp]
    lib.icv_threshold_pack.argtypes = [vp, P(Matrix), vp, vp, dbl, i32, vp, i64, vp, vp, i64, i64, vp, vp, vp, i64, vp]
    lib.icv_row_offsets.argtypes = [vp, i64, vp, vp]
    lib.icv_pack_geometry.argtypes = [i32, P(PackInfo)]
    lib.icv_csr_fill_masked.argtypes = [vp, i64, i32, i64, vp, vp, vp, vp, vp]
    lib.icv_corr_iqr.argtypes = [vp, i64, i32, i64, P(C.c_double), vp]
    lib.icv_pairwise_sqeuclidean.argtypes = [vp, i64, i32, i64, i64, i64, vp, i64, vp]
    lib.icv_ward_linkage.argtypes = [vp, i64, i64, i32, vp, P(i32), vp]
    lib.icv_pairwise_sqeuclidean_tiles.argtypes = [vp, i64, i32, i64, i32, vp, vp, vp, vp, vp, i64, vp, i64, vp]
    lib.icv_ward_create.argtypes = [i64, vp, i32, i32, i64, i32, P(vp), vp]
    lib.icv_ward_destroy.argtypes = [vp]
    lib.icv_ward_destroy.restype = None
    lib.icv_ward_merge.argtypes = [vp, vp, i64, vp, i64, vp, i32, vp]
    lib.icv_ward_gather.argtypes = [vp, vp, i64, vp, i32, vp, i32, vp, i64, vp]
    lib.icv_ward_scatter.argtypes = [vp, vp, i64, vp, i64, vp, i32, vp]
    lib.icv_ward_scan.argtypes = [vp, vp, i64, vp]
    lib.icv_ward_pack_nn.argtypes = [vp, vp, vp, vp]
    lib.icv_ward_unpack_nn.argtypes = [vp, vp, vp, vp]
    lib.icv_ward_pairs.argtypes = [vp, vp, i64, i32, P(i32), vp]
    lib.icv_ward_round_pairs.argtypes = [vp, vp, vp]
    lib.icv_ward_finish.argtypes = [vp, vp, P(i32)]
    lib.icv_row_abs_sum.argtypes = [vp, i64, i32, i64, vp, vp]
    lib.icv_csr_row_abs_sum.argtypes = [vp, i32, vp, i64, vp, vp]
    lib.icv_group_sums.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    lib.icv_csr_check.argtypes = [vp, vp, i64, i32, i64, vp]
    lib.icv_csr_densify.argtypes = [vp, i32, vp, vp, vp, i64, i32, vp, i64, vp]
    lib.icv_host_dense_row_nnz.argtypes = [vp, i32, i64, i64, i64, vp, i32]
    lib.icv_host_dense_pack.argtypes = [vp, i32, i64, i64, i64, vp, vp, vp, i32]
    lib.icv_host_dense_pack_fused.argtypes = [vp, i32, i64, i64, i64, vp, vp, vp, i64, i32, vp]
    lib.icv_csr_scatter_dense.argtypes = [vp, i32, vp, vp, i64, i32, vp, i64, vp]
    lib.icv_developer_knobs_reload.restype = None
    lib.icv_developer_knobs_reload.argtypes = []
    lib.icv_last_error.restype = C.c_char_p
    lib.icv_last_error.argtypes = []
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("icv_plan_destroy", "icv_ward_destroy", "icv_last_error", "icv_developer_knobs_reload"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int):
    """Map an icv_status to the exception type the reference's Python code raises."""
    if rc == ICV_OK:
        return
    msg = load().icv_last_error().decode("utf-8", "replace")
    if rc == ICV_ERR_INVALID:
        raise ValueError(msg)
    if rc == ICV_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ICV_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def device_count() -> int:
    return int(load().icv_device_count())

"""Device-side execution of the hot path through the C ABI.

PyTorch is used only as plumbing: HBM allocation, H2D/D2H copies and streams.  Every numeric
step runs in ``libinfercnv_hip.so``.  There is no CPU fallback: without a GPU (or without the
built library) the functions here raise.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time as _time

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._plan import GenePlan


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("infercnvpy_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch


def _stream_ptr(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()


class DeviceMatrix:
    """A cells x genes matrix resident in HBM (dense row-major or CSR), float32 or float64.

    ``DeviceMatrix(dense=tensor)`` or ``DeviceMatrix(indptr=int64, indices=int32, data=float32 / float64,
    shape=(cells, genes))`` (CUDA tensors).  A CSR matrix built by the caller is checked on the device (``validate``,
    one pass over the column indices and one flag read back): row offsets non-decreasing and inside the buffers, column
    indices inside ``[0, genes)``, ascending and unique within every row -- what the host path gets from scipy's
    canonical format.  ``ValueError`` otherwise (unsorted rows would give silently wrong windows)."""

    def __init__(self, *, dense=None, indptr=None, indices=None, data=None, shape=None, indptr_host=None,
                 validate=True):
        torch = _torch()
        if dense is not None:
            if not (isinstance(dense, torch.Tensor) and dense.is_cuda and dense.dim() == 2):
                raise ValueError("DeviceMatrix: `dense` must be a 2-D CUDA tensor")
            if dense.dtype not in (torch.float32, torch.float64):
                raise ValueError("DeviceMatrix: a device matrix must be float32 or float64")
            if dense.stride(1) != 1 and dense.shape[1] > 1:
                raise ValueError("DeviceMatrix: rows must be contiguous (stride 1 along the genes)")
            self.format = _lib.ICV_DENSE
            self.dense = dense
            self.shape = tuple(dense.shape)
            self.dtype = dense.dtype
            self._keep = (dense,)
        else:
            for name, t in (("indptr", indptr), ("indices", indices), ("data", data)):
                if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 1 and t.is_contiguous()):
                    raise ValueError(f"DeviceMatrix: `{name}` must be a contiguous 1-D CUDA tensor")
            if shape is None or len(tuple(shape)) != 2:
                raise ValueError("DeviceMatrix: a CSR matrix needs shape=(cells, genes)")
            if indptr.dtype != torch.int64 or indices.dtype != torch.int32:
                raise ValueError("DeviceMatrix: indptr must be int64 and indices int32")
            if data.dtype not in (torch.float32, torch.float64):
                raise ValueError("DeviceMatrix: a device matrix must be float32 or float64")
            if indptr.numel() != int(shape[0]) + 1:
                raise ValueError("DeviceMatrix: indptr must hold cells + 1 offsets")
            if not (indptr.device == indices.device == data.device):
                raise ValueError("DeviceMatrix: indptr, indices and data must live on one GPU")
            self.format = _lib.ICV_CSR
            self.indptr, self.indices, self.data = indptr, indices, data
            self.shape = (int(shape[0]), int(shape[1]))
            self.dtype = data.dtype
            self._keep = (indptr, indices, data)
            if validate:
                with torch.cuda.device(indptr.device):
                    _lib.check(_lib.load().icv_csr_check(_ptr(indptr), _ptr(indices), self.shape[0], self.shape[1],
                                                         min(indices.numel(), data.numel()), _stream_ptr(torch)))
            # host copy of the row pointers (8 B per row): slices need indptr[row0], indptr[row1]
            self.indptr_host = indptr_host if indptr_host is not None else indptr.cpu().numpy()

    @property
    def device(self):
        return self._keep[0].device

    def row_len_hint(self):
        """CSR: the 99.9 % quantile of the row lengths (computed once from the host copy of the row pointers): what the
        stored-entries kernel sizes its per-cell entry slots with; 0 for dense matrices."""
        if self.format != _lib.ICV_CSR:
            return 0
        hint = getattr(self, "_row_len_hint", None)
        if hint is None:
            lens = np.diff(np.asarray(self.indptr_host))
            hint = int(np.partition(lens, max(0, int(0.999 * (lens.shape[0] - 1))))[max(0, int(0.999 * (lens.shape[0] - 1)))]) \
                if lens.shape[0] else 0
            hint = int(min(max(hint, 0), 2 ** 31 - 1))
            self._row_len_hint = hint
        return hint

    def c_struct(self, row0=0, row1=None):
        """icv_matrix for rows [row0, row1)."""
        torch = _torch()
        n = self.shape[0]
        row1 = n if row1 is None else row1
        m = _lib.Matrix()
        m.format = self.format
        m.dtype = _lib.ICV_F32 if self.dtype == torch.float32 else _lib.ICV_F64
        m.n_rows = row1 - row0
        m.n_cols = self.shape[1]
        esz = 4 if self.dtype == torch.float32 else 8
        if self.format == _lib.ICV_DENSE:
            m.ld = self.dense.stride(0)
            m.values = self.dense.data_ptr() + row0 * self.dense.stride(0) * esz
        else:
            m.ld = 0
            m.values = self.data.data_ptr()
            m.indptr = self.indptr.data_ptr() + 8 * row0  # absolute offsets into indices/values
            m.indices = self.indices.data_ptr()
            m.csr_begin = int(self.indptr_host[row0])
            m.csr_end = int(self.indptr_host[row1])
            m._pad = self.row_len_hint()  # (performance hint of k_smooth_se: the length most rows stay under)
        return m


def to_device_matrix(X, dtype=None, device="cuda"):
    """numpy ndarray / scipy sparse -> DeviceMatrix (host data are copied once)."""
    torch = _torch()
    if isinstance(X, DeviceMatrix):
        return X
    if isinstance(X, torch.Tensor):
        if dtype is not None and X.dtype != dtype:
            X = X.to(dtype)
        return DeviceMatrix(dense=X.to(device).contiguous())
    np_dtype = {None: None, torch.float32: np.float32, torch.float64: np.float64}[dtype]
    if sp.issparse(X):
        X = X.tocsr()
        if not X.has_canonical_format:  # the kernels expect unique, sorted column indices per row
            X = X.copy()
            X.sum_duplicates()
        data = X.data if np_dtype is None else X.data.astype(np_dtype, copy=False)
        indptr64 = X.indptr.astype(np.int64, copy=False)
        return DeviceMatrix(
            validate=False,  # canonical scipy CSR
            indptr_host=indptr64,
            indptr=torch.from_numpy(np.ascontiguousarray(indptr64)).to(device),
            indices=torch.from_numpy(np.ascontiguousarray(X.indices.astype(np.int32, copy=False))).to(device),
            data=torch.from_numpy(np.ascontiguousarray(data)).to(device),
            shape=X.shape,
        )
    if isinstance(X, np.matrix):
        X = np.asarray(X)
    X = np.ascontiguousarray(X if np_dtype is None else X.astype(np_dtype, copy=False))
    return DeviceMatrix(dense=torch.from_numpy(X).to(device))


def column_sums(dm: DeviceMatrix, row_group=None, n_groups=1, sums=None, row0=0, row1=None):
    """float64 per-group column sums of rows [row0, row1), accumulated into ``sums`` (device n_groups x n_cols);
    ``row_group``: group of every one of those rows (-1 = none)."""
    torch = _torch()
    lib = _lib.load()
    if sums is None:
        sums = torch.zeros((n_groups, dm.shape[1]), dtype=torch.float64, device="cuda")
    rg = None
    if row_group is not None:
        rg = torch.as_tensor(np.asarray(row_group, dtype=np.int32)).to("cuda")
    m = dm.c_struct(row0, row1)
    _lib.check(lib.icv_colsum(C.byref(m), _ptr(rg), n_groups, _ptr(sums), _stream_ptr(torch)))
    return sums


def column_chain(dm: DeviceMatrix, acc=None, rows=None, count=None, row0=0, row1=None, cols=None):
    """Continue the reference-order column sums (``icv_colchain``): ``acc`` (device, ``n_cols`` values of the matrix
    dtype; None = start from zero) += the rows ``rows`` (ascending indices RELATIVE to ``row0``; None = all) of
    ``dm[row0:row1]``, added one after the other as numpy's ``np.mean(X, axis=0)`` / scipy's CSR ``mean`` add them.
    ``count``: rows of the WHOLE group (CSR input multiplies every entry by ``1 / count`` first, as scipy does).
    ``cols = (c0, c1)`` (dense only): only the columns [c0, c1) -- ``acc`` stays the full-width vector, its slice is
    continued (the chains of different columns are independent: ranks pipeline over column groups, ``dist.py``)."""
    torch = _torch()
    lib = _lib.load()
    if acc is None:
        acc = torch.zeros(dm.shape[1], dtype=dm.dtype, device="cuda")
    assert acc.dtype == dm.dtype and acc.is_cuda and acc.numel() == dm.shape[1] and acc.is_contiguous()
    m = dm.c_struct(row0, row1)
    acc_ptr = acc.data_ptr()
    if cols is not None:
        c0, c1 = int(cols[0]), int(cols[1])
        if dm.format != _lib.ICV_DENSE:
            raise ValueError("column_chain: column ranges are for dense matrices")
        if not 0 <= c0 <= c1 <= dm.shape[1]:
            raise ValueError("column_chain: column range out of bounds")
        if c1 == c0:
            return acc
        esz = 4 if dm.dtype == torch.float32 else 8
        m.values += c0 * esz
        m.n_cols = c1 - c0
        m._pad = c0  # column offset of the view inside its row (the library's bounds reasoning needs it)
        acc_ptr += c0 * esz
    rows_d, n_sel = None, m.n_rows
    if rows is not None:
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        n_sel = int(rows.shape[0])
        if n_sel == 0:
            return acc
        rows_d = torch.from_numpy(rows).to("cuda")
    scale = 0.0
    if dm.format == _lib.ICV_CSR:
        if not count:
            raise ValueError("column_chain: CSR input needs the row count of the group")
        scale = 1.0 / float(count)
    _lib.check(lib.icv_colchain(C.byref(m), _ptr(rows_d), n_sel, scale, C.c_void_p(acc_ptr), _stream_ptr(torch)))
    return acc


class ChainBlocks:
    """The reference-order float32 chain of a dense float32 matrix BY BLOCKS of rows (``icv_colchain_blocks_*``,
    csrc/icv_kernel_blocks.hpp): what row-sharded ranks run so that they do not take turns on the chain.

        cb = ChainBlocks(dm)                 # workspace on the device
        total = cb.sums()                    # float64 column totals of these rows (concurrent; all-gather them)
        cb.records(est_start)                # block records from the float64 estimate of the chain before row 0
        cb.scan(acc)                         # acc: EXACT float32 chain values before row 0 -> after the last row

    ``scan`` equals ``column_chain(dm, acc)`` bit for bit whatever ``est_start`` was (a wrong estimate costs replays)."""

    def __init__(self, dm: "DeviceMatrix", row0=0, row1=None):
        torch = _torch()
        lib = _lib.load()
        self.dm = dm
        self.m = dm.c_struct(row0, row1)
        nbytes = C.c_int64(0)
        _lib.check(lib.icv_colchain_blocks_workspace(self.m.n_rows, self.m.n_cols, C.byref(nbytes)))
        self.ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device="cuda")
        self.replayed = torch.zeros(1, dtype=torch.int64, device="cuda")

    def sums(self):
        torch = _torch()
        total = torch.empty(self.m.n_cols, dtype=torch.float64, device="cuda")
        _lib.check(_lib.load().icv_colchain_blocks_sums(C.byref(self.m), _ptr(self.ws), _ptr(total), _stream_ptr(torch)))
        return total

    def records(self, est_start=None):
        torch = _torch()
        if est_start is not None:
            assert est_start.dtype == torch.float64 and est_start.is_cuda and est_start.numel() == self.m.n_cols
            est_start = est_start.contiguous()
        _lib.check(_lib.load().icv_colchain_blocks_records(C.byref(self.m), _ptr(self.ws), _ptr(est_start),
                                                           _stream_ptr(torch)))

    def scan(self, acc, cols=None):
        torch = _torch()
        assert acc.dtype == torch.float32 and acc.is_cuda and acc.numel() == self.m.n_cols and acc.is_contiguous()
        c0, c1 = (0, self.m.n_cols) if cols is None else (int(cols[0]), int(cols[1]))
        _lib.check(_lib.load().icv_colchain_blocks_scan(C.byref(self.m), _ptr(self.ws), _ptr(acc), c0, c1,
                                                        _ptr(self.replayed), _stream_ptr(torch)))
        return acc

    def n_blocks(self):
        return (-(-int(self.m.n_rows) // 64)) * int(self.m.n_cols)


def chain_mean(acc, count, is_csr):
    """Means from the accumulators of :func:`column_chain`: ``acc / count`` in the matrix dtype for dense input
    (numpy's ``true_divide(sum, n)``); CSR accumulators already are the means (scipy scales the entries)."""
    if is_csr:
        return acc
    torch = _torch()
    lib = _lib.load()
    out = torch.empty_like(acc)
    _lib.check(lib.icv_colchain_mean(_ptr(acc), _lib.ICV_F32 if acc.dtype == torch.float32 else _lib.ICV_F64,
                                     acc.numel(), int(count), _ptr(out), _stream_ptr(torch)))
    return out


def csc_column_means(X, row_group=None, n_groups=1, counts=None, np_dtype=None, max_entries=1 << 28):
    """Per-group column means of a host scipy CSC matrix in scipy's own order (``np.add.reduceat`` per column: first
    stored entry + numpy's pairwise sum of the others, entries scaled by 1 / n first): ``n_groups x n_cols`` host array
    of ``np_dtype``.  The columns travel to the GPU in blocks of at most ``max_entries`` stored entries."""
    torch = _torch()
    lib = _lib.load()
    X = X.tocsc()
    if not X.has_canonical_format:
        X = X.copy()
        X.sum_duplicates()
    np_dtype = np.dtype(np_dtype or (np.float32 if X.dtype == np.float32 else np.float64))
    tdt = torch.float32 if np_dtype == np.float32 else torch.float64
    code = _lib.ICV_F32 if np_dtype == np.float32 else _lib.ICV_F64
    n_cols = X.shape[1]
    indptr = X.indptr.astype(np.int64, copy=False)
    out = np.zeros((n_groups, n_cols), dtype=np_dtype)
    rg = None
    if row_group is not None:
        rg = torch.from_numpy(np.ascontiguousarray(row_group, dtype=np.int32)).cuda()
    if counts is None:
        counts = [X.shape[0]] * n_groups
    c = 0
    while c < n_cols:
        c1 = int(np.searchsorted(indptr, indptr[c] + max_entries, side="right")) - 1
        c1 = min(n_cols, max(c1, c + 1))
        e0, e1 = int(indptr[c]), int(indptr[c1])
        vals = torch.from_numpy(np.ascontiguousarray(X.data[e0:e1].astype(np_dtype, copy=False))).cuda()
        rows = torch.from_numpy(np.ascontiguousarray(X.indices[e0:e1].astype(np.int32, copy=False))).cuda()
        ptr = torch.from_numpy(np.ascontiguousarray(indptr[c:c1 + 1] - e0)).cuda()
        if e1 == e0:  # nothing stored: (the kernel never dereferences empty arrays, but give it valid pointers)
            vals = torch.zeros(1, dtype=tdt, device="cuda")
            rows = torch.zeros(1, dtype=torch.int32, device="cuda")
        for g in range(n_groups):
            mean = torch.empty(c1 - c, dtype=tdt, device="cuda")
            _lib.check(lib.icv_colmean_csc(_ptr(vals), code, _ptr(ptr), _ptr(rows), c1 - c, _ptr(rg), g,
                                           1.0 / float(counts[g]), _ptr(mean), _stream_ptr(torch)))
            out[g, c:c1] = mean.cpu().numpy()
        c = c1
    return out


def fortran_column_means(X, np_dtype=None, max_bytes=1 << 30):
    """All-cell column means of a dense HOST matrix stored column-major (``X.strides[0] < X.strides[1]``: an
    F-ordered array, the transposed view of a genes x cells matrix) in numpy's own order: numpy reduces such a matrix
    column by column with its contiguous inner loop -- pairwise summation over pieces of 8 192 elements -- where a
    C-ordered matrix gets one sequential chain per column (reference tl/_infercnv.py:385).  The columns travel to the GPU
    in blocks as they lie in host memory (``icv_colsum_pairwise``); host array of ``np_dtype``."""
    torch = _torch()
    lib = _lib.load()
    n, g = X.shape
    np_dtype = np.dtype(np_dtype or (np.float32 if X.dtype == np.float32 else np.float64))
    tdt = torch.float32 if np_dtype == np.float32 else torch.float64
    code = _lib.ICV_F32 if np_dtype == np.float32 else _lib.ICV_F64
    XT = X.T
    out = np.empty(g, dtype=np_dtype)
    per = int(max(1, max_bytes // max(1, n * np_dtype.itemsize)))
    for c0 in range(0, g, per):
        c1 = min(g, c0 + per)
        blk = torch.from_numpy(np.ascontiguousarray(XT[c0:c1].astype(np_dtype, copy=False))).cuda()
        sums = torch.empty(c1 - c0, dtype=tdt, device="cuda")
        _lib.check(lib.icv_colsum_pairwise(_ptr(blk), code, n, c1 - c0, n, _ptr(sums), _stream_ptr(torch)))
        out[c0:c1] = chain_mean(sums, n, False).cpu().numpy()
    return out


def free_hbm_bytes():
    """HBM this process can still allocate on the current device: what the driver reports free PLUS the blocks PyTorch's
    caching allocator holds without using them (it hands those out again, or gives them back before it reports out of
    memory) -- after a previous large call the driver's figure alone makes the next call cut its matrix into needless
    slabs (a 1 M-cell matrix was uploaded twice for the reference pass and the smoothing)."""
    torch = _torch()
    free_b, _ = torch.cuda.mem_get_info()
    return int(free_b) + max(0, int(torch.cuda.memory_reserved()) - int(torch.cuda.memory_allocated()))


def alloc_out(rows, n_windows):
    """Device float32 ``rows x n_windows`` result buffer whose rows start on 16-byte boundaries (row stride padded
    to a multiple of 4): the smoothing kernel then writes x_res with 16-byte stores."""
    torch = _torch()
    ld = (n_windows + 3) // 4 * 4
    return torch.empty((rows, ld), dtype=torch.float32, device="cuda")[:, :n_windows]


class SmoothResult:
    windows = None  # float64 windows before centring (run_hot_path(windows=True))

    def __init__(self, out, cell_median, cell_stats, thr, profile):
        self.out = out                  # device float32 C x W (thresholded x_res)
        self.cell_median = cell_median  # device float64 C
        self.cell_stats = cell_stats    # device float64 C x 2 | None (not requested)
        self.thr = thr                  # device float64 n_chunks | None
        self.profile = profile


def _usable_cpus():
    """Logical CPUs this process may really use: the affinity mask, capped by a cgroup CPU quota when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def _default_pack_threads():
    """Host threads that pack a dense piece (ICV_PACK_THREADS overrides): three quarters of the usable logical CPUs
    (affinity mask and cgroup quota), at most 32.  Measured on the GPU pool's box (2 x 64 cores, but a cgroup quota of 16 CPUs:
    tools/bench_host_pack.py, profiles/r05_host_pack.txt): 124 GB/s of input with 8 threads, 162 with 16, 173 with 32,
    then a collapse (46 GB/s with 64, 20 with 128: the threads are throttled and wait on each other's block tickets)."""
    env = os.environ.get("ICV_PACK_THREADS")
    if env:
        return max(1, int(env))
    # three quarters of the usable CPUs: the packing is bound by host memory, not by the cores (12 threads pack as fast
    # as 16 on a 16-CPU quota: 0.107 / 0.109 s per 16 GB), and the copier, the drain and the page-fault helpers run beside
    # it -- a process that asks for more CPU time than its cgroup grants is stopped for the rest of the 100 ms period
    # (the 0.19 s repeats between 0.14 s ones in profiles/r06_bench_n1.json; 24 / 32 threads: 0.16-0.26 s per call)
    n = _usable_cpus()
    return int(max(1, min(32, n - n // 4)))


# Host buffers of the sparse upload, kept between calls (first-touch page faults of a few GB of fresh pages cost as much as
# the packing itself): at most _PACK_POOL_MAX buffer pairs and _PACK_POOL_BYTES of host memory (six 1 GB pairs
# stayed allocated after the call), handed out largest first; release_pinned_buffers() empties it.
_PACK_POOL = []
_PACK_POOL_LOCK = None
_PACK_POOL_MAX = 6
_PACK_POOL_BYTES = 3 << 30


def _pack_pool_take(np_dtype):
    import threading

    global _PACK_POOL_LOCK
    if _PACK_POOL_LOCK is None:
        _PACK_POOL_LOCK = threading.Lock()
    with _PACK_POOL_LOCK:
        best = None
        for k, b in enumerate(_PACK_POOL):
            if b[1].dtype == np_dtype and (best is None or b[0].shape[0] > _PACK_POOL[best][0].shape[0]):
                best = k
        if best is not None:
            return _PACK_POOL.pop(best)
    return [np.empty(0, np.int32), np.empty(0, np_dtype)]


def _pack_pool_give(bufs):
    if _PACK_POOL_LOCK is None or bufs[0].shape[0] == 0:
        return
    with _PACK_POOL_LOCK:
        held = sum(b[0].nbytes + b[1].nbytes for b in _PACK_POOL)
        if len(_PACK_POOL) < _PACK_POOL_MAX and held + bufs[0].nbytes + bufs[1].nbytes <= _PACK_POOL_BYTES:
            _PACK_POOL.append(bufs)


def _probe_density(X, np_dtype, row0, row1, probe_rows=256):
    """Stored entries per element in ``probe_rows`` rows of X[row0:row1] (a C-ordered float matrix) taken in 16 runs
    spread evenly over the range (a matrix sorted by library size -- empty droplets first -- is not what its
    first rows look like)."""
    n_all = row1 - row0
    n = min(probe_rows, n_all)
    runs = 16 if n_all >= 16 * 16 else 1
    per = max(1, n // runs)
    lib = _lib.load()
    code = _lib.ICV_F32 if np_dtype == np.float32 else _lib.ICV_F64
    total = rows = 0
    for k in range(runs):
        a = row0 + (k * (n_all - per)) // max(runs - 1, 1) if runs > 1 else row0
        xs = X[a:a + per]
        cnt = np.zeros(xs.shape[0], dtype=np.int64)
        _lib.check(lib.icv_host_dense_row_nnz(xs.ctypes.data, code, xs.shape[0], X.shape[1], X.strides[0] // X.itemsize,
                                              cnt.ctypes.data, 1))
        total += int(cnt.sum())
        rows += xs.shape[0]
    return float(total) / float(max(1, rows * X.shape[1]))


def _wants_sparse_upload(X, np_dtype, row0, row1, max_density=0.3):
    """Dense host rows go up as stored entries when they are mostly zeros: a C-ordered float matrix of the compute
    dtype, at least 1024 columns, fewer than ``max_density`` non-zeros in a probe of the first rows.
    ``ICV_NO_SPARSE_UPLOAD`` switches it off (A/B timing)."""
    if os.environ.get("ICV_NO_SPARSE_UPLOAD"):
        return False
    if not isinstance(X, np.ndarray) or X.ndim != 2 or X.dtype != np_dtype or X.shape[1] < 1024 or row1 <= row0:
        return False
    if X.strides[1] != X.itemsize or X.strides[0] < X.shape[1] * X.itemsize or X.strides[0] % X.itemsize:
        return False
    return _probe_density(X, np_dtype, row0, row1) < max_density


class SlabStream:
    """Upload the rows of a host matrix piece by piece on a side stream while the caller computes.

    A helper thread issues the host -> HBM copies (pageable source: the copy call blocks its thread, not the
    caller; measured 56 GB/s on PCIe Gen5 x16, the same as from pinned memory) and records an event per piece;
    ``pieces()`` yields ``(r0, r1)`` in order once the current stream has been made to wait for that piece.
    ``dm`` is the DeviceMatrix of the whole slab (valid row ranges: the pieces yielded so far).
    """

    def __init__(self, X, tdtype, piece_rows, row0=0, row1=None, host_pack_threads=None):
        """Rows [row0, row1) of the host matrix ``X`` (the parent's arrays are read in place: slicing a scipy CSR
        matrix would copy the shard -- 5.6 GB at BASELINE config 4 -- before the first byte is uploaded).
        ``host_pack_threads``: host threads of the sparse upload of a mostly-zero dense matrix (default: the usable
        logical CPUs, at most 32; callers with several shards divide them)."""
        import queue
        import threading

        torch = _torch()
        np_dtype = np.float32 if tdtype == torch.float32 else np.float64
        row1 = X.shape[0] if row1 is None else row1
        self.n_rows = row1 - row0
        self.bounds = [(r, min(self.n_rows, r + piece_rows)) for r in range(0, self.n_rows, piece_rows)]
        self._landed = 0
        device = torch.cuda.current_device()  # the helper threads start on device 0 otherwise
        creator_stream = torch.cuda.current_stream()
        self._err = None
        self._cancel = threading.Event()
        self.sparse_upload = False
        self._packer = None
        # one uploader thread (and stream) per host array: a CSR slab is two arrays, and two threads keep the link
        # busy where one thread alternating between them does not (measured with cold pages and a device -> host
        # copy beside them: 54 against 39 GB/s, tools/exp_h2d_csr.py)
        if sp.issparse(X):
            base = int(X.indptr[row0])  # the slab's entries are [base, base + nnz) of the parent's arrays
            indptr64 = np.ascontiguousarray(X.indptr[row0:row1 + 1].astype(np.int64)) - base
            nnz = int(indptr64[-1])
            d_indices = torch.empty(max(nnz, 1), dtype=torch.int32, device="cuda")
            d_data = torch.empty(max(nnz, 1), dtype=tdtype, device="cuda")
            d_indptr = torch.from_numpy(indptr64).cuda()
            self.dm = DeviceMatrix(indptr=d_indptr, indices=d_indices, data=d_data, shape=(self.n_rows, X.shape[1]),
                                   indptr_host=indptr64, validate=False)  # (filled later, from a canonical matrix)
            idx_h, dat_h = X.indices, X.data

            def copy_indices(r0, r1):
                k0, k1 = int(indptr64[r0]), int(indptr64[r1])
                if k1 > k0:
                    d_indices[k0:k1].copy_(torch.from_numpy(
                        np.ascontiguousarray(idx_h[base + k0:base + k1].astype(np.int32, copy=False))))

            def copy_data(r0, r1):
                k0, k1 = int(indptr64[r0]), int(indptr64[r1])
                if k1 > k0:
                    d_data[k0:k1].copy_(torch.from_numpy(
                        np.ascontiguousarray(dat_h[base + k0:base + k1].astype(np_dtype, copy=False))))

            copiers = [copy_indices, copy_data]
        else:
            dense = torch.empty((self.n_rows, X.shape[1]), dtype=tdtype, device="cuda")
            self.dm = DeviceMatrix(dense=dense)
            self.sparse_upload = _wants_sparse_upload(X, np_dtype, row0, row1)

            def copy_piece(r0, r1):
                dense[r0:r1].copy_(torch.from_numpy(
                    np.ascontiguousarray(X[row0 + r0:row0 + r1].astype(np_dtype, copy=False))))

            if self.sparse_upload:
                # A dense matrix of log-counts is ~80 % zeros and PCIe is what the call waits for: a packer thread turns
                # every piece into (indptr, indices, values) on host threads (icv_host_dense_pack: 8 bytes per stored
                # entry instead of 4 per element), the copier thread sends the three arrays and rebuilds the dense rows
                # in HBM (icv_csr_scatter_dense) -- the slab is bit for bit what the dense copy would have delivered.
                lib = _lib.load()
                code = _lib.ICV_F32 if np_dtype == np.float32 else _lib.ICV_F64
                n_cols = X.shape[1]
                ld = X.strides[0] // X.itemsize
                n_thr = host_pack_threads if host_pack_threads else _default_pack_threads()
                packed_q = queue.Queue(maxsize=2)  # packed pieces waiting for the link
                free_q = queue.Queue()             # recycled host buffers: their pages stay mapped, within the call and
                for _ in range(3):                 # (through the process-wide pool) between calls
                    free_q.put(_pack_pool_take(np_dtype))
                self._pack_bufs = free_q
                self.pack_stats = {"pack_s": 0.0, "wait_buffer_s": 0.0, "wait_link_s": 0.0, "threads": n_thr}
                stats = self.pack_stats
                dens = [max(_probe_density(X, np_dtype, row0, row1), 1e-4)]  # stored entries per element (estimate)

                def pack_all():
                    try:
                        for r0, r1 in self.bounds:
                            if self._cancel.is_set():
                                break
                            xs = X[row0 + r0:row0 + r1]
                            ip = np.empty(r1 - r0 + 1, dtype=np.int64)
                            t1 = _time.perf_counter()
                            bufs = None
                            while bufs is None:  # (a copier that has failed or been cancelled returns nothing)
                                try:
                                    bufs = free_q.get(timeout=0.1)
                                except queue.Empty:
                                    if self._cancel.is_set() or self._err is not None:
                                        return
                            want = int(dens[0] * 1.25 * (r1 - r0) * n_cols) + 4096
                            if bufs[0].shape[0] < want:
                                bufs[0], bufs[1] = np.empty(want, np.int32), np.empty(want, np_dtype)
                            t2 = _time.perf_counter()
                            # one pass over the piece (icv_host_dense_pack_fused); more entries than the buffers hold
                            # (the probe under-estimated the density): grow them and pack again
                            nnz_c = C.c_int64(0)
                            for _ in range(2):
                                rc = lib.icv_host_dense_pack_fused(xs.ctypes.data, code, r1 - r0, n_cols, ld, ip.ctypes.data,
                                                                   bufs[0].ctypes.data, bufs[1].ctypes.data,
                                                                   bufs[0].shape[0], n_thr, C.byref(nnz_c))
                                if rc != _lib.ICV_ERR_NOMEM or nnz_c.value <= bufs[0].shape[0]:
                                    break
                                cap = int(nnz_c.value * 1.1) + 4096
                                bufs[0], bufs[1] = np.empty(cap, np.int32), np.empty(cap, np_dtype)
                                dens[0] = max(dens[0], nnz_c.value / float(max(1, (r1 - r0) * n_cols)))
                            _lib.check(rc)
                            nnz = int(nnz_c.value)
                            t3 = _time.perf_counter()
                            while True:
                                try:
                                    packed_q.put((ip, nnz, bufs), timeout=0.1)
                                    break
                                except queue.Full:
                                    if self._cancel.is_set() or self._err is not None:
                                        return
                            stats["wait_buffer_s"] += t2 - t1
                            stats["pack_s"] += t3 - t2
                            stats["wait_link_s"] += _time.perf_counter() - t3
                    except BaseException as e:  # surfaced by the copier
                        try:
                            packed_q.put(e, timeout=5)
                        except queue.Full:
                            self._err = e

                def copy_piece(r0, r1):  # noqa: F811 -- the sparse form replaces the dense copy
                    # (polled: a packer that was cancelled before it started this piece enqueues nothing -- ADVICE r5)
                    item = None
                    while item is None:
                        try:
                            item = packed_q.get(timeout=0.1)
                        except queue.Empty:
                            if self._err is not None:
                                raise self._err
                            if self._cancel.is_set():
                                raise RuntimeError("sparse upload cancelled")
                            if not self._packer.is_alive() and packed_q.empty():
                                raise RuntimeError("sparse upload: the packer stopped without delivering the piece")
                    if isinstance(item, BaseException):
                        raise item
                    ip, nnz, bufs = item
                    d_ip = torch.from_numpy(ip).cuda()
                    d_idx = torch.empty(max(nnz, 1), dtype=torch.int32, device="cuda")
                    d_val = torch.empty(max(nnz, 1), dtype=tdtype, device="cuda")
                    if nnz:
                        d_idx[:nnz].copy_(torch.from_numpy(bufs[0][:nnz]))
                        d_val[:nnz].copy_(torch.from_numpy(bufs[1][:nnz]))
                    free_q.put(bufs)  # (pageable copies have left the host buffers when copy_ returns)
                    _lib.check(lib.icv_csr_scatter_dense(_ptr(d_val), code, _ptr(d_ip), _ptr(d_idx), r1 - r0, n_cols,
                                                         C.c_void_p(dense[r0:r1].data_ptr()), dense.stride(0),
                                                         _stream_ptr(torch)))

                self._packer = threading.Thread(target=pack_all, daemon=True)

            copiers = [copy_piece]

        self._streams = [torch.cuda.Stream() for _ in copiers]
        self._queues = [queue.Queue() for _ in copiers]
        self._busy = [0.0 for _ in copiers]

        def work(k):
            import time

            t0 = time.perf_counter()
            stream, q, copy = self._streams[k], self._queues[k], copiers[k]
            try:
                torch.cuda.set_device(device)
                # the buffers were allocated on the creator's stream: whatever that stream still has queued on a
                # recycled block (kernels of the previous slab) comes before the first copy into it
                stream.wait_stream(creator_stream)
                with torch.cuda.stream(stream):
                    for r0, r1 in self.bounds:
                        if self._cancel.is_set():
                            break
                        copy(r0, r1)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        q.put(ev)
                stream.synchronize()
            except BaseException as e:  # surfaced in the consumer
                self._err = e
                q.put(None)
            self._busy[k] = time.perf_counter() - t0

        self._threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(len(copiers))]
        if getattr(self, "_packer", None) is not None:
            self._threads.append(self._packer)
        for th in self._threads:
            th.start()

    @property
    def h2d_seconds(self):
        """Wall-clock seconds the slowest uploader thread took (they run side by side)."""
        return max(self._busy)

    def pieces(self):
        """Row ranges in order; blocks (stream-wise) only for pieces that have not landed yet.  Re-iterable."""
        torch = _torch()
        for k in range(len(self.bounds)):
            if k >= self._landed:
                for q in self._queues:  # a piece has landed when every array's part of it has
                    ev = q.get()
                    if ev is None:
                        raise self._err
                    torch.cuda.current_stream().wait_event(ev)
                self._landed += 1
            yield self.bounds[k]
        self._join()

    def _join(self):
        for th in self._threads:
            if th.is_alive():
                th.join()

    def close(self, cancel=False):
        """Stop the uploader (``cancel``: skip the pieces it has not started) and drop the slab's device buffers."""
        if cancel:
            self._cancel.set()
        self._join()
        bufs_q = getattr(self, "_pack_bufs", None)
        if bufs_q is not None:  # the host buffers of the sparse upload go back to the process-wide pool
            self._pack_bufs = None
            while True:
                try:
                    _pack_pool_give(bufs_q.get_nowait())
                except Exception:
                    break
        if self.dm is not None:
            # kernels of the consumer's stream may still read the slab: the caching allocator must not hand the
            # blocks (allocated on the creator's stream) to anyone before those kernels have run
            torch = _torch()
            cur = torch.cuda.current_stream()
            for t in self.dm._keep:
                t.record_stream(cur)
                for st in self._streams:
                    t.record_stream(st)
        self.dm = None


class PackedRows:
    """x_res of a row range with its keep-mask and per-row counts (device), awaiting the CSR fill."""

    def __init__(self, out, mask, counts, thr):
        self.out, self.mask, self.counts, self.thr = out, mask, counts, thr


def threshold_mask(plan: GenePlan, dm: DeviceMatrix, ref_lo, ref_hi, res, *, lfc_clip, chunksize, row_phase=0,
                   flags=0, row0=0, row1=None):
    """Step 5b without rewriting x_res: keep-bits (|x| >= thr decided in float64, x != 0) + per-row counts."""
    torch = _torch()
    lib = _lib.load()
    rows = res.out.shape[0]
    n_words = (plan.n_windows + 63) // 64
    mask = torch.empty((rows, n_words), dtype=torch.int64, device="cuda")
    counts = torch.empty(rows, dtype=torch.int64, device="cuda")
    m = dm.c_struct(row0, row1)
    _lib.check(lib.icv_threshold_mask(
        plan.handle, C.byref(m), _ptr(ref_lo), _ptr(ref_hi), float(lfc_clip), int(flags), _ptr(res.out),
        res.out.stride(0), _ptr(res.cell_median), _ptr(res.thr), int(chunksize), int(row_phase), _ptr(mask),
        _ptr(counts), _stream_ptr(torch)))
    return PackedRows(res.out, mask, counts, res.thr)


class PackedCsr:
    """X_cnv of a row range as device CSR: ``indptr`` (rows + 1 int64, from 0), ``indices`` (int32) / ``data`` (float64)
    buffers of which the first ``indptr[-1]`` entries are valid.

    This is what ``tl.infercnv`` leaves in ``obsm["X_cnv"]`` for an HBM-resident matrix; ``tl.cnv_score``,
    ``tl.ithcna``, ``tl.cell_linkage`` / ``tl.ward_linkage`` and ``pl.chromosome_heatmap(_summary)`` take it as they
    take the host matrix (the tool functions work on the device arrays; only the plots copy it to the host)."""

    def __init__(self, indptr, indices, data, n_cols, thr=None):
        self.indptr, self.indices, self.data, self.n_cols, self.thr = indptr, indices, data, n_cols, thr

    @property
    def n_rows(self):
        return self.indptr.shape[0] - 1

    @property
    def shape(self):
        return (self.n_rows, self.n_cols)

    @property
    def device(self):
        return self.indptr.device

    def nnz(self):
        """Number of stored entries (reads one value back: synchronises the current stream)."""
        return int(self.indptr[-1].item())

    def to_scipy(self):
        """Host ``scipy.sparse.csr_matrix`` (float64), as the reference returns."""
        ip = self.indptr.cpu().numpy()
        n = int(ip[-1])
        return sp.csr_matrix((self.data[:n].cpu().numpy(), self.indices[:n].cpu().numpy(), ip),
                             shape=(self.n_rows, self.n_cols))

    def toarray(self):
        """Host dense float64 matrix (``scipy``'s ``toarray``)."""
        return self.to_scipy().toarray()

    def dense_rows(self, rows=None):
        """Device float32 ``len(rows) x n_cols`` tile of the selected rows (boolean mask or index array; None = all):
        the input of the fp32 MFMA contractions (``icv_csr_densify``); the values are float32 numbers widened to
        float64, so nothing is rounded.  Row stride padded to 16 bytes."""
        torch = _torch()
        lib = _lib.load()
        with torch.cuda.device(self.device):
            rows_d, n_sel = None, self.n_rows
            if rows is not None:
                rows = np.asarray(rows)
                if rows.dtype == bool:
                    if rows.shape[0] != self.n_rows:
                        raise IndexError("boolean row mask of the wrong length")
                    rows = np.flatnonzero(rows)
                rows = np.ascontiguousarray(rows, dtype=np.int64)
                if rows.size and (rows.min() < 0 or rows.max() >= self.n_rows):
                    raise IndexError("row index out of range")
                n_sel = int(rows.shape[0])
                rows_d = torch.from_numpy(rows).cuda()
            ld = (self.n_cols + 3) // 4 * 4
            out = torch.empty((n_sel, max(ld, 1)), dtype=torch.float32, device="cuda")[:, :self.n_cols]
            code = _lib.ICV_F32 if self.data.dtype == torch.float32 else _lib.ICV_F64
            _lib.check(lib.icv_csr_densify(_ptr(self.data), code, _ptr(self.indptr), _ptr(self.indices), _ptr(rows_d),
                                           n_sel, self.n_cols, _ptr(out), out.stride(0), _stream_ptr(torch)))
        return out


def threshold_csr(plan: GenePlan, dm: DeviceMatrix, ref_lo, ref_hi, res, *, lfc_clip, chunksize, row_phase=0,
                  flags=0, row0=0, row1=None, capacity=None, single_pass=False):
    """Step 5b + ``csr_matrix(x_res)`` (reference :449-455) on the device without a host round trip: the un-thresholded
    ``res.out`` of :func:`run_hot_path` (``apply=False``) -> :class:`PackedCsr`.  ``capacity`` (entries) defaults to the
    worst case rows x windows; nothing is read back, the call is asynchronous.

    Default: keep-mask + row counts (``icv_threshold_mask``: ``k_thr_mask_ring``, x_res streamed through an LDS ring),
    ``icv_row_offsets``, ``icv_csr_fill_masked`` (``k_csr_fill_ring``) -- 0.33 ms per 100 000 cells.
    ``single_pass=True``: ``icv_threshold_pack`` (one kernel with a decoupled look-back; 0.7 ms, kept for comparison --
    profiles/r04_pack_experiments.txt)."""
    torch = _torch()
    lib = _lib.load()
    rows = res.out.shape[0]
    W = plan.n_windows
    if (dm.shape[0] if row1 is None else row1) - row0 == 0:  # an empty shard packs to an empty matrix
        return PackedCsr(torch.zeros(1, dtype=torch.int64, device="cuda"), torch.empty(1, dtype=torch.int32, device="cuda"),
                         torch.empty(1, dtype=torch.float64, device="cuda"), W, res.thr)
    if single_pass and W <= 20480:
        return threshold_pack(plan, dm, ref_lo, ref_hi, res, lfc_clip=lfc_clip, chunksize=chunksize,
                              row_phase=row_phase, flags=flags, row0=row0, row1=row1, capacity=capacity)
    part = threshold_mask(plan, dm, ref_lo, ref_hi, res, lfc_clip=lfc_clip, chunksize=chunksize, row_phase=row_phase,
                          flags=flags, row0=row0, row1=row1)
    # (the fill kernels write at the row offsets without a bound: never less than the worst case here; a smaller
    # ``capacity`` is honoured by the single-pass form only, which drops the overflow)
    cap = rows * W if capacity is None else max(int(capacity), rows * W)
    indptr = torch.empty(rows + 1, dtype=torch.int64, device="cuda")
    indices = torch.empty(max(cap, 1), dtype=torch.int32, device="cuda")
    data = torch.empty(max(cap, 1), dtype=torch.float64, device="cuda")
    st = _stream_ptr(torch)
    _lib.check(lib.icv_row_offsets(_ptr(part.counts), rows, _ptr(indptr), st))
    if rows:
        _lib.check(lib.icv_csr_fill_masked(_ptr(part.out), rows, W, part.out.stride(0), _ptr(part.mask), _ptr(indptr),
                                           _ptr(indices), _ptr(data), st))
    return PackedCsr(indptr, indices, data, W, res.thr)


def threshold_pack(plan: GenePlan, dm: DeviceMatrix, ref_lo, ref_hi, res, *, lfc_clip, chunksize, row_phase=0,
                   flags=0, row0=0, row1=None, capacity=None):
    """The single-pass form (``icv_threshold_pack``, at most 20 480 windows): see :func:`threshold_csr`."""
    torch = _torch()
    lib = _lib.load()
    rows = res.out.shape[0]
    W = plan.n_windows
    cap = rows * W if capacity is None else int(capacity)
    indptr = torch.empty(rows + 1, dtype=torch.int64, device="cuda")
    indices = torch.empty(max(cap, 1), dtype=torch.int32, device="cuda")
    data = torch.empty(max(cap, 1), dtype=torch.float64, device="cuda")
    m = dm.c_struct(row0, row1)
    _lib.check(lib.icv_threshold_pack(
        plan.handle, C.byref(m), _ptr(ref_lo), _ptr(ref_hi), float(lfc_clip), int(flags), _ptr(res.out),
        res.out.stride(0), _ptr(res.cell_median), _ptr(res.thr), int(chunksize), int(row_phase), _ptr(indptr),
        _ptr(indices), _ptr(data), cap, _stream_ptr(torch)))
    return PackedCsr(indptr, indices, data, W, res.thr)


def fill_from_mask(part, n_cols):
    """Two-step packing of a :class:`PackedRows` keep-mask (window lists too long for ``icv_threshold_pack``):
    prefix sum, read-back of the count, ``icv_csr_fill_masked`` -> :class:`PackedCsr`."""
    torch = _torch()
    lib = _lib.load()
    n = part.counts.shape[0]
    ip = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(part.counts, 0, out=ip[1:])
    nnz = int(ip[-1].item())
    idx = torch.empty(max(nnz, 1), dtype=torch.int32, device="cuda")
    dat = torch.empty(max(nnz, 1), dtype=torch.float64, device="cuda")
    if nnz:
        _lib.check(lib.icv_csr_fill_masked(_ptr(part.out), n, n_cols, part.out.stride(0), _ptr(part.mask), _ptr(ip),
                                           _ptr(idx), _ptr(dat), _stream_ptr(torch)))
    return PackedCsr(ip, idx, dat, n_cols, part.thr)


def concat_packed(parts):
    """Row-wise concatenation of device CSR pieces (reads each piece's entry count back)."""
    torch = _torch()
    ips, idx, dat, thr, off = [parts[0].indptr[:1]], [], [], [], 0
    for p in parts:
        n = p.nnz()
        ips.append(p.indptr[1:] + off)
        idx.append(p.indices[:n])
        dat.append(p.data[:n])
        if p.thr is not None:
            thr.append(p.thr)
        off += n
    return PackedCsr(torch.cat(ips), torch.cat(idx), torch.cat(dat), parts[0].n_cols, torch.cat(thr) if thr else None)


_PAGE = 4096


def _prefault(arrays, start=0, n_threads=8):
    """Write one byte into every page of freshly allocated numpy arrays (from element ``start`` on), from a few
    threads in parallel: numpy releases the GIL for the strided store, the page faults run concurrently."""
    import threading

    jobs = []
    for a in arrays:
        b = a.view(np.uint8).reshape(-1)
        lo = (start * a.itemsize) // _PAGE * _PAGE
        n_pages = (b.shape[0] - lo + _PAGE - 1) // _PAGE
        if n_pages < 256:  # < 1 MB: not worth a thread
            continue
        per = (n_pages + n_threads - 1) // n_threads
        for k in range(n_threads):
            p0, p1 = k * per, min(n_pages, (k + 1) * per)
            if p0 < p1:
                jobs.append(b[lo + p0 * _PAGE: min(b.shape[0], lo + p1 * _PAGE): _PAGE])
    threads = [threading.Thread(target=lambda v=v: v.fill(0)) for v in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()


class _PinnedRing:
    """Device -> pageable host copies through a small ring of pinned staging buffers (one ring per GPU and process).

    A copy straight into pageable memory runs at ~13 GB/s through the driver's staging path and holds back the
    uploads running beside it; here the DMA engine writes 16 MB chunks into pinned slots at link speed and a few
    host threads move them into the final arrays (numpy copies release the GIL)."""

    _rings = {}
    _lock = None

    @classmethod
    def get(cls, torch):
        import atexit
        import threading

        if cls._lock is None:
            cls._lock = threading.Lock()
            atexit.register(cls.release_all)
        dev = torch.cuda.current_device()
        with cls._lock:
            if dev not in cls._rings:
                cls._rings[dev] = cls(torch)
            return cls._rings[dev]

    @classmethod
    def release_all(cls):
        """Drop every GPU's ring (128 MB of pinned host memory and three helper threads each); the next drain builds a
        new one.  Registered with atexit; ``infercnvpy_amd._engine.release_pinned_buffers()`` for long-lived processes."""
        if cls._lock is None:
            return
        with cls._lock:
            rings, cls._rings = cls._rings, {}
        for r in rings.values():
            r.pool.shutdown(wait=True)
            r.slots, r.views = [], []

    def __init__(self, torch, n_slots=8, slot_bytes=16 << 20, n_threads=3):
        import queue
        from concurrent.futures import ThreadPoolExecutor

        self.torch = torch
        self.slot_bytes = slot_bytes
        self.slots = [torch.empty(slot_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(n_slots)]
        self.views = [t.numpy() for t in self.slots]
        self.free = queue.Queue()
        for i in range(n_slots):
            self.free.put(i)
        self.pool = ThreadPoolExecutor(n_threads, thread_name_prefix="icv-d2h")

    def _land(self, i, ev, dst, n):
        try:
            ev.synchronize()
            np.copyto(dst, self.views[i][:n])
        finally:
            self.free.put(i)

    def download(self, src, dst, stream):
        """Enqueue ``dst[...] = src`` (1-D device tensor -> contiguous numpy array of the same dtype and length) on
        ``stream``; returns the futures of the chunks (the copy is complete when all of them are)."""
        torch = self.torch
        src_b = src.view(torch.uint8)
        dst_b = dst.view(np.uint8).reshape(-1)
        n_bytes = src_b.numel()
        assert dst_b.shape[0] == n_bytes
        futs = []
        for a in range(0, n_bytes, self.slot_bytes):
            n = min(self.slot_bytes, n_bytes - a)
            i = self.free.get()
            with torch.cuda.stream(stream):
                self.slots[i][:n].copy_(src_b[a:a + n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
            futs.append(self.pool.submit(self._land, i, ev, dst_b[a:a + n], n))
        return futs


def release_pinned_buffers():
    """Free the pinned staging rings of the device -> host copies (128 MB per GPU used so far, kept between calls) and the
    host buffers of the sparse upload."""
    _PinnedRing.release_all()
    if _PACK_POOL_LOCK is not None:
        with _PACK_POOL_LOCK:
            del _PACK_POOL[:]
    else:
        del _PACK_POOL[:]


class CsrDrain:
    """CSR pack + copy back of finished pieces on a side stream while the caller computes the next ones.

    ``submit(part)`` (a :class:`PackedCsr` -- the piece packed on the device by ``icv_threshold_pack`` -- or a
    :class:`PackedRows` mask for the two-step form; consecutive row ranges in order) returns at once, or blocks while
    ``max_pending`` pieces are still queued (a packed piece holds a worst-case buffer).  A helper thread waits for the
    piece on its own stream, reads the row offsets back and copies the packed entries into the final host arrays (sized
    from the first piece's density, grown if that was too small).  ``finish()`` returns the scipy CSR matrix of all
    rows."""

    def __init__(self, n_rows, n_cols, max_pending=2):
        import queue
        import threading

        torch = _torch()
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self._q = queue.Queue()
        self._slots = threading.Semaphore(max_pending)
        self._stream = torch.cuda.Stream()
        self._err = None
        self._cancel = threading.Event()
        self._finished = False
        self.indptr_h = np.zeros(self.n_rows + 1, dtype=np.int64)
        self.indices_h = np.empty(0, dtype=np.int32)
        self.data_h = np.empty(0, dtype=np.float64)
        self.rows_done, self.nnz = 0, 0
        self.busy_seconds = 0.0
        device = torch.cuda.current_device()
        lib = _lib.load()
        ring = None if os.environ.get("ICV_NO_PINNED_D2H") else _PinnedRing.get(torch)  # knob: A/B timing
        self._pending = []  # futures of chunks on their way into indices_h / data_h
        settle = self._settle

        def reserve(extra, rows_after):
            need = self.nnz + extra
            if need <= self.indices_h.shape[0]:
                return
            settle()  # the arrays are about to be replaced
            # density so far extrapolated to all rows, + 25 %
            est = int(need / max(rows_after, 1) * self.n_rows * 1.25) + 1024 if rows_after < self.n_rows else need
            cap = max(need, est)
            # (plain allocations: huge-page-advised mappings were measured and are 2x slower to fill from the device)
            idx, dat = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.float64)
            # Fresh pages are mapped (and zeroed) by the kernel at first touch.  Left to the device -> host copy, those
            # faults happen inside the driver's copy path, one at a time, and slow the uploads running beside it as
            # well (measured: 39 instead of 55 GB/s host -> HBM).  Touch them here from a few threads instead.
            if not os.environ.get("ICV_NO_PREFAULT"):  # developer knob for A/B timing
                _prefault((idx, dat), start=self.nnz)
            idx[: self.nnz] = self.indices_h[: self.nnz]
            dat[: self.nnz] = self.data_h[: self.nnz]
            self.indices_h, self.data_h = idx, dat

        def work():
            import time

            try:
                torch.cuda.set_device(device)
                with torch.cuda.stream(self._stream):
                    while True:
                        item = self._q.get()
                        if item is None:
                            return
                        if self._cancel.is_set():  # a failed call: drop what is queued, keep draining to the sentinel
                            del item
                            self._slots.release()
                            continue
                        part, ev = item
                        t0 = time.perf_counter()
                        self._stream.wait_event(ev)
                        if isinstance(part, PackedCsr):
                            n = part.n_rows
                            ip, idx_d, dat_d = part.indptr, part.indices, part.data
                            ip_h = ip.cpu().numpy()
                            nnz = int(ip_h[-1])
                            if nnz > idx_d.shape[0]:
                                raise RuntimeError(f"icv_threshold_pack: {nnz} entries for a capacity of {idx_d.shape[0]}")
                            held = (ip, idx_d, dat_d)
                        else:
                            n = part.counts.shape[0]
                            ip = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
                            torch.cumsum(part.counts, 0, out=ip[1:])
                            ip_h = ip.cpu().numpy()
                            nnz = int(ip_h[-1])
                            idx_d = dat_d = None
                            held = (part.out, part.mask, part.counts)
                            if nnz:
                                idx_d = torch.empty(nnz, dtype=torch.int32, device="cuda")
                                dat_d = torch.empty(nnz, dtype=torch.float64, device="cuda")
                                _lib.check(lib.icv_csr_fill_masked(
                                    _ptr(part.out), n, self.n_cols, part.out.stride(0), _ptr(part.mask), _ptr(ip),
                                    _ptr(idx_d), _ptr(dat_d), self._stream.cuda_stream))
                        reserve(nnz, self.rows_done + n)
                        if nnz:
                            o = self.nnz
                            if ring is not None:
                                self._pending += ring.download(idx_d[:nnz], self.indices_h[o:o + nnz], self._stream)
                                self._pending += ring.download(dat_d[:nnz], self.data_h[o:o + nnz], self._stream)
                            else:
                                torch.from_numpy(self.indices_h[o:o + nnz]).copy_(idx_d[:nnz])
                                torch.from_numpy(self.data_h[o:o + nnz]).copy_(dat_d[:nnz])
                        # the piece's buffers were allocated on the CONSUMER's stream and are read here on the drain's
                        # (fill kernel, asynchronous copies): the caching allocator must not hand them out before this
                        # stream is done with them
                        for t in held:
                            if t is not None:
                                t.record_stream(self._stream)
                        del idx_d, dat_d, held
                        self.indptr_h[self.rows_done + 1:self.rows_done + n + 1] = ip_h[1:] + self.nnz
                        self.rows_done += n
                        self.nnz += nnz
                        del part, item, ip
                        self._slots.release()
                        self.busy_seconds += time.perf_counter() - t0
            except BaseException as e:  # surfaced in finish()
                self._err = e
                try:
                    settle()
                except BaseException:
                    pass
                while True:  # keep consuming (and releasing) submitted parts until the sentinel arrives
                    try:
                        if self._q.get(timeout=60) is None:
                            return
                        self._slots.release()
                    except queue.Empty:
                        return

        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def _settle(self):
        # (a method, not a closure kept in an attribute: that was a reference cycle, and the drain -- with the host arrays
        # of X_cnv, 3.5 GB at 1 M cells -- stayed alive until the cyclic collector happened to run, inside someone's
        # timed region: profiles/r06_gc_stall.txt)
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()

    def submit(self, part):
        torch = _torch()
        while not self._slots.acquire(timeout=0.05):  # back-pressure: at most max_pending pieces wait for the drain
            if self._err is not None:
                raise self._err
        if self._err is not None:
            raise self._err
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._q.put((part, ev))

    def finish(self, arrays=False):
        """Wait for the queued pieces; the scipy CSR matrix of all rows, or with ``arrays=True`` its
        ``(indptr, indices, data)`` (the caller concatenates several drains)."""
        self._finished = True
        self._q.put(None)
        self._thread.join()
        if self._err is not None:
            raise self._err
        self._settle()
        assert self.rows_done == self.n_rows, (self.rows_done, self.n_rows)
        if arrays:
            return self.indptr_h, self.indices_h[: self.nnz], self.data_h[: self.nnz]
        return sp.csr_matrix((self.data_h[: self.nnz], self.indices_h[: self.nnz], self.indptr_h),
                             shape=(self.n_rows, self.n_cols))

    def close(self):
        """Idempotent shutdown for error paths: drop queued pieces (their device buffers), stop the thread."""
        if not self._finished:
            self._finished = True
            self._cancel.set()
            self._q.put(None)
        if self._thread.is_alive():
            self._thread.join()
        try:
            self._settle()
        except BaseException:
            pass


def run_hot_path(plan: GenePlan, dm: DeviceMatrix, ref_lo, ref_hi=None, *, lfc_clip=3.0, dynamic_threshold=1.5,
                 chunksize=5000, row_phase=0, flags=0, out=None, profile=False, row0=0, row1=None,
                 cell_stats=False, apply=True, windows=False):
    """Steps 1-5 of the chunk kernel for rows [row0, row1) of ``dm`` (all device-resident).

    ``cell_stats=True`` also returns the per-cell moments (sum, sum of squares of x_res); without them the
    library forms the noise-threshold moments per chunk inside the smoothing kernel (faster).
    ``apply=False`` computes the thresholds but leaves ``out`` un-thresholded (for :func:`threshold_mask`).
    ``windows=True``: the float64 windows before the per-cell centring come back as well (``res.windows``, rows x W:
    what ``calculate_gene_values`` averages, :func:`gene_values_from_windows`) -- written by the same kernel launch."""
    torch = _torch()
    lib = _lib.load()
    n = dm.shape[0]
    row1 = n if row1 is None else row1
    rows = row1 - row0
    W = plan.n_windows
    if out is None:
        out = alloc_out(rows, W)
    assert out.dtype == torch.float32 and out.shape[0] >= rows and out.shape[1] >= W and out.stride(1) == 1
    med = torch.empty(rows, dtype=torch.float64, device="cuda")
    stats = torch.empty((rows, 2), dtype=torch.float64, device="cuda") if cell_stats else None
    if rows == 0:  # a rank that owns no rows (more ranks than chunks): nothing to launch, no chunk, no threshold
        res = SmoothResult(out, med, stats,
                           None if dynamic_threshold is None else torch.empty(0, dtype=torch.float64, device="cuda"), None)
        res.windows = torch.empty((0, W), dtype=torch.float64, device="cuda") if windows else None
        return res
    dyn = float("nan") if dynamic_threshold is None else float(dynamic_threshold)
    thr = None
    if dynamic_threshold is not None:
        n_chunks = max(1, math.ceil((rows + row_phase) / chunksize))
        thr = torch.empty(n_chunks, dtype=torch.float64, device="cuda")
    assert ref_lo.dtype == dm.dtype and ref_lo.is_cuda and ref_lo.numel() == dm.shape[1]
    if ref_hi is not None:
        assert ref_hi.dtype == dm.dtype and ref_hi.is_cuda and ref_hi.numel() == dm.shape[1]
    m = dm.c_struct(row0, row1)
    prof = _lib.Profile() if profile else None
    win = torch.empty((rows, W), dtype=torch.float64, device="cuda") if windows else None
    _lib.check(lib.icv_infercnv_run_windows(
        plan.handle, C.byref(m), _ptr(ref_lo), _ptr(ref_hi), float(lfc_clip), dyn, int(chunksize), int(row_phase),
        int(flags) | (0 if apply else _lib.ICV_FLAG_NO_APPLY), _ptr(out), out.stride(0), _ptr(med), _ptr(stats),
        _ptr(thr),
        C.byref(prof) if prof is not None else None, _ptr(win), W, _stream_ptr(torch)))
    res = SmoothResult(out, med, stats, thr, prof)
    res.windows = win
    return res


def row_abs_sum(x_cnv):
    """per-row sum |x| of a device float32 matrix (cnv_score building block)."""
    torch = _torch()
    lib = _lib.load()
    assert x_cnv.is_cuda and x_cnv.dtype == torch.float32 and x_cnv.stride(1) == 1
    res = torch.empty(x_cnv.shape[0], dtype=torch.float64, device="cuda")
    _lib.check(lib.icv_row_abs_sum(_ptr(x_cnv), x_cnv.shape[0], x_cnv.shape[1], x_cnv.stride(0), _ptr(res),
                                   _stream_ptr(torch)))
    return res


def csr_row_abs_sum(x_csr):
    """per-row sum |x| of a CSR X_cnv (device float64 vector): a host scipy matrix sends only indptr and the stored
    values to the GPU, a :class:`PackedCsr` is read where it lies."""
    torch = _torch()
    lib = _lib.load()
    if isinstance(x_csr, PackedCsr):
        with torch.cuda.device(x_csr.device):
            res = torch.empty(x_csr.n_rows, dtype=torch.float64, device="cuda")
            code = _lib.ICV_F32 if x_csr.data.dtype == torch.float32 else _lib.ICV_F64
            _lib.check(lib.icv_csr_row_abs_sum(_ptr(x_csr.data), code, _ptr(x_csr.indptr), x_csr.n_rows, _ptr(res),
                                               _stream_ptr(torch)))
        return res
    x_csr = x_csr.tocsr()
    data = x_csr.data if x_csr.data.dtype in (np.float32, np.float64) else x_csr.data.astype(np.float64)
    d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    ip = torch.from_numpy(np.ascontiguousarray(x_csr.indptr.astype(np.int64, copy=False))).cuda()
    res = torch.empty(x_csr.shape[0], dtype=torch.float64, device="cuda")
    _lib.check(lib.icv_csr_row_abs_sum(_ptr(d), _lib.ICV_F32 if data.dtype == np.float32 else _lib.ICV_F64, _ptr(ip),
                                       x_csr.shape[0], _ptr(res), _stream_ptr(torch)))
    return res


def group_sums(values, codes, n_groups, want_counts=True):
    """(sums, counts) per group of a device float64 vector: ``codes`` (host int32, -1 = no group) names the group of
    every element.  Fixed summation order on the device (``icv_group_sums``); host numpy arrays of length n_groups
    (``want_counts=False``: counts is None -- one read-back instead of two)."""
    torch = _torch()
    lib = _lib.load()
    with torch.cuda.device(values.device):
        codes_d = torch.from_numpy(np.ascontiguousarray(codes, dtype=np.int32)).cuda()
        sums = torch.empty(max(n_groups, 1), dtype=torch.float64, device="cuda")
        counts = torch.empty(max(n_groups, 1), dtype=torch.int64, device="cuda")
        _lib.check(lib.icv_group_sums(_ptr(values), _ptr(codes_d), values.numel(), int(n_groups), _ptr(sums),
                                      _ptr(counts), _stream_ptr(torch)))
        return sums[:n_groups].cpu().numpy(), (counts[:n_groups].cpu().numpy() if want_counts else None)


def gene_values(plan: GenePlan, dm: DeviceMatrix, ref_lo, ref_hi=None, *, lfc_clip=3.0, thr=None, chunksize=5000,
                row_phase=0, flags=0, row0=0, row1=None, out=None):
    """calculate_gene_values for rows [row0, row1): float64 ``rows x n_vars`` device tensor (``out``: written in place),
    NaN where a gene has no value.  ``thr``: the thresholds of the chunks of THOSE rows."""
    torch = _torch()
    lib = _lib.load()
    row1 = dm.shape[0] if row1 is None else row1
    if out is None:
        out = torch.empty((row1 - row0, dm.shape[1]), dtype=torch.float64, device="cuda")
    assert out.dtype == torch.float64 and out.shape[0] == row1 - row0 and out.shape[1] == dm.shape[1]
    assert out.stride(1) == 1 or out.shape[1] <= 1
    m = dm.c_struct(row0, row1)
    _lib.check(lib.icv_gene_values(
        plan.handle, C.byref(m), _ptr(ref_lo), _ptr(ref_hi), float(lfc_clip), int(flags), _ptr(thr), int(chunksize),
        int(row_phase), _ptr(out), out.stride(0), _stream_ptr(torch)))
    return out


def gene_values_from_windows(plan: GenePlan, windows, *, thr=None, chunksize=5000, row_phase=0, n_vars=None, out=None):
    """calculate_gene_values from the float64 windows of :func:`run_hot_path` (``windows=True``): float64
    ``rows x n_vars`` device tensor (``out``: written in place, completely), NaN where a gene has no value.  ONE kernel
    (``icv_gene_values_from_windows``); ``thr``: the thresholds of the chunks of those rows."""
    torch = _torch()
    lib = _lib.load()
    rows = windows.shape[0]
    assert windows.dtype == torch.float64 and windows.is_cuda and (windows.stride(1) == 1 or windows.shape[1] <= 1)
    if out is None:
        out = torch.empty((rows, int(n_vars)), dtype=torch.float64, device="cuda")
    assert out.dtype == torch.float64 and out.shape[0] == rows and (out.stride(1) == 1 or out.shape[1] <= 1)
    _lib.check(lib.icv_gene_values_from_windows(
        plan.handle, _ptr(windows), windows.stride(0), rows, _ptr(thr), int(chunksize), int(row_phase), _ptr(out),
        out.stride(0), _stream_ptr(torch)))
    return out


def dense_to_host_csr(out, n_cols):
    """Dense float32 device result -> scipy CSR float64 on the host, packed on the GPU first."""
    torch = _torch()
    lib = _lib.load()
    rows = out.shape[0]
    counts = torch.empty(rows, dtype=torch.int64, device="cuda")
    _lib.check(lib.icv_csr_count(_ptr(out), rows, n_cols, out.stride(0), _ptr(counts), _stream_ptr(torch)))
    indptr = torch.zeros(rows + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(counts, 0, out=indptr[1:])
    nnz = int(indptr[-1].item())
    indices = torch.empty(max(nnz, 1), dtype=torch.int32, device="cuda")
    data = torch.empty(max(nnz, 1), dtype=torch.float64, device="cuda")
    _lib.check(lib.icv_csr_fill(_ptr(out), rows, n_cols, out.stride(0), _ptr(indptr), _ptr(indices), _ptr(data),
                                _stream_ptr(torch)))
    return sp.csr_matrix((data[:nnz].cpu().numpy(), indices[:nnz].cpu().numpy(), indptr.cpu().numpy()),
                         shape=(rows, n_cols))


def corr_iqr(x):
    """IQR of all entries of np.corrcoef(x) for a device float32 matrix (cells x features)."""
    torch = _torch()
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    out = C.c_double()
    _lib.check(lib.icv_corr_iqr(_ptr(x), x.shape[0], x.shape[1], x.stride(0), C.byref(out), _stream_ptr(torch)))
    return float(out.value)


def spare_stride(n):
    """Row stride (floats) of an n x n distance matrix with the n / 2 spare columns the Ward rounds can use."""
    return (n + (n + 1) // 2 + 3) // 4 * 4


def has_spare_columns(dist_sq):
    n = dist_sq.shape[0]
    return dist_sq.stride(0) % 4 == 0 and dist_sq.stride(0) >= n + (n + 1) // 2


def pairwise_sqeuclidean(x, out=None, rows=None, spare=False):
    """float32 squared Euclidean distances between the rows of a device matrix: the full n x n matrix, or the
    row block ``rows = (begin, end)`` against all n rows.  ``spare=True`` (full matrix, no ``out``): allocate the
    result with n / 2 spare columns per row when HBM allows (6 n^2 bytes instead of 4 n^2) -- for a matrix that goes
    on to :func:`ward_linkage` (``spare=has_spare_columns(out)``), whose rounds then write their column updates as
    dense strips."""
    torch = _torch()
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    n = x.shape[0]
    r0, r1 = (0, n) if rows is None else rows
    if out is None:
        ld = (n + 3) // 4 * 4  # a multiple of 16 bytes: vector loads in the Ward rounds
        if rows is None and spare:
            free_b = free_hbm_bytes()
            if 4 * n * spare_stride(n) + 4 * n * (x.shape[1] + 16) + (1 << 30) < free_b:
                ld = spare_stride(n)
        out = torch.empty((r1 - r0, ld), dtype=torch.float32, device=x.device)[:, :n]
    assert out.shape[0] >= r1 - r0 and out.shape[1] >= n and out.stride(1) == 1
    _lib.check(lib.icv_pairwise_sqeuclidean(_ptr(x), n, x.shape[1], x.stride(0), r0, r1, _ptr(out), out.stride(0),
                                            _stream_ptr(torch)))
    return out


def ward_linkage(dist_sq, spare=False):
    """scipy-format Ward linkage matrix from a device n x n squared-distance matrix (overwritten).  ``spare=True``:
    the columns [n, stride) of every row are the rounds' to use as well (a matrix allocated with
    :func:`spare_stride`); False: nothing outside the n x n block is written, whatever the stride."""
    torch = _torch()
    lib = _lib.load()
    assert dist_sq.is_cuda and dist_sq.dtype == torch.float32 and dist_sq.dim() == 2 and dist_sq.stride(1) == 1
    n = dist_sq.shape[0]
    assert dist_sq.shape[1] == n
    Z = np.empty((max(n - 1, 0), 4), dtype=np.float64)
    rounds = C.c_int32(0)
    _lib.check(lib.icv_ward_linkage(_ptr(dist_sq), n, dist_sq.stride(0), 1 if spare else 0, Z.ctypes.data,
                                    C.byref(rounds), _stream_ptr(torch)))
    return Z, int(rounds.value)


def profile_begin(plan: GenePlan):
    """Start deferred timing: later run_hot_path calls on this plan record HIP events without synchronising."""
    _lib.check(_lib.load().icv_profile_begin(plan.handle))


def profile_collect(plan: GenePlan, max_records=4096):
    """End deferred timing; returns the list of _lib.Profile records of the runs since profile_begin."""
    arr = (_lib.Profile * max_records)()
    n = C.c_int32(0)
    _lib.check(_lib.load().icv_profile_collect(plan.handle, arr, max_records, C.byref(n)))
    return [arr[i] for i in range(n.value)]

"""CPU tests of the HOST functions behind the sparse upload of a mostly-zero dense matrix (no GPU needed: they are plain
multi-threaded C++ in the C-ABI library) and of the host-side rules around them."""
import ctypes as C
import os

import numpy as np
import pytest

from infercnvpy_amd import _lib


def _expected(M):
    bits = np.ascontiguousarray(M).view(np.uint32 if M.dtype == np.float32 else np.uint64)
    mask = bits != 0  # the BIT PATTERN decides: NaN and -0.0 are stored entries
    return mask.sum(axis=1), np.nonzero(mask)[1].astype(np.int32), bits[mask]


def _cases(dt):
    rs = np.random.RandomState(0)
    X = rs.gamma(0.3, 1.0, (1203, 2001)).astype(dt)
    X[X < 0.5] = 0
    X[5] = 1.5          # a full row
    X[7] = 0            # an empty row
    X[9, 3] = np.nan
    X[10, 4] = -0.0
    X[11, :-3] = 2.0    # nearly full
    wide = np.zeros((1203, 2048), dt)
    wide[:, :2001] = X
    return [X, wide[:, :2001], X[:, :15], X[:, :16], X[:, :33], X[:5], X[:16], X[:17], X[:0]]


@pytest.mark.parametrize("dt,code", [(np.float32, _lib.ICV_F32), (np.float64, _lib.ICV_F64)])
def test_two_pass_and_one_pass_packing_equal_the_definition(dt, code):
    """(AVX-512 forms where this CPU has them -- the dispatch is decided once per process; the scalar forms run in
    test_scalar_forms_in_a_child_process.)"""
    lib = _lib.load()
    for M in _cases(dt):
        n, g = M.shape
        ld = M.strides[0] // M.itemsize if n else g
        cnt, idx_e, val_e = _expected(M) if n else (np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.uint32))
        for thr in (1, 3, 16):
            ip = np.zeros(n + 1, np.int64)
            if n:
                _lib.check(lib.icv_host_dense_row_nnz(M.ctypes.data, code, n, g, ld, ip[1:].ctypes.data, thr))
                np.testing.assert_array_equal(ip[1:], cnt)
            np.cumsum(ip, out=ip)
            nnz = int(ip[-1])
            idx = np.full(nnz + 3, -7, np.int32)
            val = np.full(nnz + 3, -7, dt)
            if n:
                _lib.check(lib.icv_host_dense_pack(M.ctypes.data, code, n, g, ld, ip.ctypes.data, idx.ctypes.data,
                                                   val.ctypes.data, thr))
                np.testing.assert_array_equal(idx[:nnz], idx_e)
                np.testing.assert_array_equal(val[:nnz].view(val_e.dtype), val_e)
                assert (idx[nnz:] == -7).all() and (val[nnz:] == -7).all()  # nothing written past the last entry
            # one pass
            ip2 = np.full(n + 1, -1, np.int64)
            idx2 = np.full(nnz + 3, -7, np.int32)
            val2 = np.full(nnz + 3, -7, dt)
            got = C.c_int64(-1)
            _lib.check(lib.icv_host_dense_pack_fused(M.ctypes.data if n else idx2.ctypes.data, code, n, g, ld,
                                                     ip2.ctypes.data, idx2.ctypes.data, val2.ctypes.data, nnz, thr,
                                                     C.byref(got)))
            assert got.value == nnz
            np.testing.assert_array_equal(ip2, ip)
            np.testing.assert_array_equal(idx2[:nnz], idx_e)
            np.testing.assert_array_equal(val2[:nnz].view(val_e.dtype), val_e)
            assert (idx2[nnz:] == -7).all()


def test_one_pass_reports_the_count_when_the_buffers_are_too_small():
    lib = _lib.load()
    X = np.random.RandomState(1).rand(300, 1100).astype(np.float32)
    X[X < 0.7] = 0
    nnz = int((X != 0).sum())
    ip = np.empty(301, np.int64)
    idx = np.full(100, -7, np.int32)
    val = np.full(100, -7, np.float32)
    got = C.c_int64(0)
    rc = lib.icv_host_dense_pack_fused(X.ctypes.data, _lib.ICV_F32, 300, 1100, 1100, ip.ctypes.data, idx.ctypes.data,
                                       val.ctypes.data, 100, 4, C.byref(got))
    assert rc == _lib.ICV_ERR_NOMEM and got.value == nnz and ip[-1] == nnz
    with pytest.raises(MemoryError):
        _lib.check(rc)
    with pytest.raises(ValueError):
        _lib.check(lib.icv_host_dense_row_nnz(X.ctypes.data, 7, 300, 1100, 1100, ip.ctypes.data, 2))


def test_scalar_forms_in_a_child_process():
    """ICV_NO_AVX512=1 (read once per process) takes the scalar packing loops: same arrays."""
    import subprocess
    import sys

    code = r"""
import ctypes as C, numpy as np
from infercnvpy_amd import _lib
lib = _lib.load()
X = np.random.RandomState(1).gamma(0.3, 1.0, (500, 3001)).astype(np.float32); X[X < 0.5] = 0
n, g = X.shape
ip = np.zeros(n + 1, np.int64)
assert lib.icv_host_dense_row_nnz(X.ctypes.data, 0, n, g, g, ip[1:].ctypes.data, 3) == 0
np.cumsum(ip, out=ip)
idx = np.empty(ip[-1], np.int32); val = np.empty(ip[-1], np.float32)
assert lib.icv_host_dense_pack(X.ctypes.data, 0, n, g, g, ip.ctypes.data, idx.ctypes.data, val.ctypes.data, 3) == 0
assert np.array_equal(idx, np.nonzero(X)[1]) and np.array_equal(val, X[X != 0])
ip2 = np.empty(n + 1, np.int64); i2 = np.empty(ip[-1], np.int32); v2 = np.empty(ip[-1], np.float32); got = C.c_int64(0)
assert lib.icv_host_dense_pack_fused(X.ctypes.data, 0, n, g, g, ip2.ctypes.data, i2.ctypes.data, v2.ctypes.data, int(ip[-1]), 3, C.byref(got)) == 0
assert np.array_equal(ip2, ip) and np.array_equal(i2, idx) and np.array_equal(v2, val)
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, ICV_NO_AVX512="1"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_usable_cpus_and_thread_default(monkeypatch):
    from infercnvpy_amd import _engine

    n = _engine._usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert 1 <= _engine._default_pack_threads() <= 32
    assert _engine._default_pack_threads() == max(1, min(32, n - n // 4))  # headroom for the copier / drain threads
    monkeypatch.setattr(_engine, "_usable_cpus", lambda: 16)
    assert _engine._default_pack_threads() == 12
    monkeypatch.setenv("ICV_PACK_THREADS", "5")
    assert _engine._default_pack_threads() == 5


def test_plan_keys_follow_the_content_of_the_annotation():
    """the resident-call plan cache was keyed by object addresses of the chromosome strings."""
    import pandas as pd

    from infercnvpy_amd.tl import _infercnv as T

    chrom = np.array(["chr1", "chr2", None, "chr1"], dtype=object)
    start = np.array([5, 1, 7, 3])
    k1 = T._plan_key(chrom, start, 100, 10, ("chrX",), 0)
    k2 = T._plan_key(np.array([str(c) + "" if c is not None else None for c in chrom], dtype=object), start.copy(), 100,
                     10, ("chrX",), 0)
    assert k1 == k2  # new string objects, same content
    chrom2 = chrom.copy()
    chrom2[0] = "chr2"
    assert T._plan_key(chrom2, start, 100, 10, ("chrX",), 0) != k1
    assert T._plan_key(chrom, start + 1, 100, 10, ("chrX",), 0) != k1
    assert T._plan_key(chrom, start, 250, 10, ("chrX",), 0) != k1
    assert T._plan_key(chrom, start, 100, 10, ("chrX",), 1) != k1
    assert T._plan_key(pd.Categorical(["chr1", "chr2", None, "chr1"]), start, 100, 10, ("chrX",), 0)[1:] == k1[1:]
    # a nullable integer column with pd.NA keys like the float column it becomes
    s_na = pd.array([5, 1, pd.NA, 3], dtype="Int64").to_numpy(dtype=object)
    assert T._plan_key(chrom, s_na, 100, 10, None, 0) == T._plan_key(chrom, np.array([5, 1, np.nan, 3]).astype(object), 100, 10, None, 0)


def test_numpys_order_for_a_column_major_matrix_restated_with_plain_loops():
    """What icv_colsum_pairwise restates (reference tl/_infercnv.py:385 on an F-ordered adata.X): numpy reduces every
    column with its contiguous inner loop -- pairwise summation (8 accumulators up to 128 elements, halves rounded down to
    multiples of 8 above) -- over pieces of 8 192 elements, the pieces added in order."""

    def pw(a):
        n = len(a)
        if n < 8:
            r = a.dtype.type(0)
            for x in a:
                r = r + x
            return r
        if n <= 128:
            r = [a[k] for k in range(8)]
            i = 8
            while i + 8 <= n:
                for k in range(8):
                    r[k] = r[k] + a[i + k]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res = res + a[i]
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return pw(a[:n2]) + pw(a[n2:])

    def numpy_order(col):
        res = None
        for s in range(0, len(col), 8192):
            p = pw(col[s:s + 8192])
            res = p if res is None else res + p
        return res

    rs = np.random.RandomState(0)
    for dt in (np.float32, np.float64):
        for n in (1, 7, 8, 129, 1000, 8192, 8193, 20000):
            X = np.asfortranarray(rs.gamma(0.3, 1.0, (n, 5)).astype(dt))
            exp = np.array([numpy_order(X[:, j]) / dt(n) for j in range(5)], dtype=dt)
            np.testing.assert_array_equal(np.mean(X, axis=0), exp)
            if n >= 129:  # and it really is another sum than the C-ordered chain
                chain = np.zeros(5, dt)
                for r in range(n):
                    chain = chain + X[r]
                np.testing.assert_array_equal(np.mean(np.ascontiguousarray(X), axis=0), chain / dt(n))

#!/usr/bin/env python
"""Host-side packing of a mostly-zero dense matrix (the sparse upload of tl.infercnv): two-pass (count, prefix sums,
pack) against the one-pass form, over thread counts.  No GPU work.  python tools/bench_host_pack.py [rows]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infercnvpy_amd import _lib  # noqa: E402

lib = _lib.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
g = 20000
rs = np.random.RandomState(0)
X = np.empty((rows, g), np.float32)
for r0 in range(0, rows, 5000):
    blk = rs.gamma(0.3, 1.0, (min(5000, rows - r0), g)).astype(np.float32)
    blk[blk < 0.5] = 0
    X[r0:r0 + blk.shape[0]] = blk
cap = int(0.25 * X.size)
idx, val = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
ip = np.zeros(rows + 1, np.int64)
print("cpus", os.cpu_count(), "matrix GB", X.nbytes / 1e9, flush=True)
for thr in (8, 16, 32, 64, 96, 128, 192):
    if thr > (os.cpu_count() or 1):
        break
    best2 = best1 = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        lib.icv_host_dense_row_nnz(X.ctypes.data, 0, rows, g, g, ip[1:].ctypes.data, thr)
        ip[0] = 0
        np.cumsum(ip, out=ip)
        t1 = time.perf_counter()
        lib.icv_host_dense_pack(X.ctypes.data, 0, rows, g, g, ip.ctypes.data, idx.ctypes.data, val.ctypes.data, thr)
        t2 = time.perf_counter()
        best2 = min(best2, t2 - t0)
        nnz = C.c_int64(0)
        t0 = time.perf_counter()
        rc = lib.icv_host_dense_pack_fused(X.ctypes.data, 0, rows, g, g, ip.ctypes.data, idx.ctypes.data, val.ctypes.data,
                                           cap, thr, C.byref(nnz))
        best1 = min(best1, time.perf_counter() - t0)
        assert rc == 0
    print(f"threads {thr:4d}  two-pass {X.nbytes / best2 / 1e9:7.1f} GB/s   one-pass {X.nbytes / best1 / 1e9:7.1f} GB/s", flush=True)

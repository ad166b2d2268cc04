"""GPU tests (``-m gpu``) of the device-side CSR packing of X_cnv: step 5b and ``csr_matrix(x_res)`` (reference
tl/_infercnv.py:449-455) as keep-mask + ``icv_row_offsets`` + fill (the public path's default) and in one pass with a
decoupled look-back over the rows (``icv_threshold_pack``), against each other, against a torch prefix sum and against
the in-place threshold: identical CSR arrays, run after run."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import cases

pytestmark = pytest.mark.gpu


def _two_step(torch, _engine, _lib, plan, dm, ref, res, chunksize, lfc_clip=3.0):
    import ctypes as C

    lib = _lib.load()
    part = _engine.threshold_mask(plan, dm, ref, None, res, lfc_clip=lfc_clip, chunksize=chunksize)
    n = part.counts.shape[0]
    ip = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(part.counts, 0, out=ip[1:])
    nnz = int(ip[-1].item())
    idx = torch.empty(max(nnz, 1), dtype=torch.int32, device="cuda")
    dat = torch.empty(max(nnz, 1), dtype=torch.float64, device="cuda")
    if nnz:
        _lib.check(lib.icv_csr_fill_masked(_engine._ptr(part.out), n, plan.n_windows, part.out.stride(0),
                                           _engine._ptr(part.mask), _engine._ptr(ip), _engine._ptr(idx),
                                           _engine._ptr(dat), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return ip.cpu().numpy(), idx[:nnz].cpu().numpy(), dat[:nnz].cpu().numpy()


@pytest.mark.parametrize("fmt,dtype,genes,window,step,dyn,n", [
    ("dense", np.float32, cases.GENES_PER_CHROM_20K, 100, 10, 1.5, 7003),
    ("csr", np.float32, cases.GENES_PER_CHROM_20K, 250, 10, 1.5, 3001),
    ("dense", np.float64, [700, 320, 150, 100, 60], 100, 10, 0.5, 1234),
    ("csr", np.float64, [700, 320, 150, 100, 60], 101, 7, None, 999),
    ("dense", np.float32, cases.GENES_PER_CHROM_20K, 100, 1, 1.5, 300),     # 17 822 windows: 279 mask words
    ("dense", np.float32, [130, 40], 100, 10, 3.0, 5000),                   # a handful of windows, most rows empty
    ("dense", np.float32, cases.GENES_PER_CHROM_20K, 100, 10, None, 1501),  # every stored window kept: the streamed fill
                                                                            # kernel takes passes; a partial last round
    ("dense", np.float32, cases.GENES_PER_CHROM_20K, 100, 5, 1.5, 803),     # 3 603 windows: 57 mask words, 19 per wavefront
    ("csr", np.float32, cases.GENES_PER_CHROM_20K, 100, 10, 0.2, 2),        # fewer rows than a round
])
def test_pack_equals_mask_and_fill(fmt, dtype, genes, window, step, dyn, n):
    from infercnvpy_amd import _engine, _lib
    from infercnvpy_amd._plan import GenePlan

    torch = _engine._torch()
    v = cases.synthetic_var(genes)
    X = cases.synthetic_expr(n, len(v["names"]), seed=n, dtype=dtype)
    ref_h = X.mean(axis=0).astype(dtype)
    dm = _engine.to_device_matrix(sp.csr_matrix(X) if fmt == "csr" else X)
    ref = torch.from_numpy(ref_h).cuda()
    plan = GenePlan(v["chromosome"], v["start"], window_size=window, step=step)
    chunksize = 500
    try:
        res = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=dyn, chunksize=chunksize, apply=False)
        ip2, ix2, dv2 = _two_step(torch, _engine, _lib, plan, dm, ref, res, chunksize)
        for rep in range(3):  # the look-back must give the same arrays whatever order the rows finish in
            pk = _engine.threshold_pack(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=chunksize)
            ip = pk.indptr.cpu().numpy()
            nnz = int(ip[-1])
            np.testing.assert_array_equal(ip, ip2)
            np.testing.assert_array_equal(pk.indices[:nnz].cpu().numpy(), ix2)
            np.testing.assert_array_equal(pk.data[:nnz].cpu().numpy(), dv2)
        # the default of the public path: mask + icv_row_offsets + fill, nothing read back
        pk = _engine.threshold_csr(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=chunksize)
        np.testing.assert_array_equal(pk.indptr.cpu().numpy(), ip2)
        np.testing.assert_array_equal(pk.indices[:nnz].cpu().numpy(), ix2)
        np.testing.assert_array_equal(pk.data[:nnz].cpu().numpy(), dv2)
        # ... and the in-place threshold of the same x_res, packed by scipy
        res_a = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=dyn, chunksize=chunksize, apply=True)
        exp = sp.csr_matrix(res_a.out.cpu().numpy().astype(np.float64))
        got = pk.to_scipy()
        assert got.shape == exp.shape
        np.testing.assert_array_equal(got.indptr, exp.indptr)
        np.testing.assert_array_equal(got.indices, exp.indices)
        np.testing.assert_array_equal(got.data, exp.data)
        # a capacity that is too small drops the overflow but still reports the count
        if nnz > 10:
            small = _engine.threshold_pack(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=chunksize, capacity=nnz // 2)
            assert small.nnz() == nnz
            np.testing.assert_array_equal(small.indices[: nnz // 2].cpu().numpy(), ix2[: nnz // 2])
    finally:
        plan.close()


@pytest.mark.parametrize("fmt,dyn,n,ties,dtype", [
    ("dense", 1.5, 4099, False, np.float32), ("csr", None, 1203, False, np.float32), ("dense", 1.5, 613, True, np.float32),
    ("csr", 1.5, 310, True, np.float32), ("dense", 1.5, 205, True, np.float64), ("csr", 1.5, 203, True, np.float64),
    ("dense", 1.5, 2051, True, "narrow"), ("dense", None, 1030, False, "narrow")])
def test_streamed_kernels_equal_the_per_row_kernels(fmt, dyn, n, ties, dtype, monkeypatch):
    """k_thr_mask_ring (+ k_thr_mask_ties) / k_csr_fill_ring against k_thr_mask / k_csr_fill_masked (developer knobs
    ICV_NO_MASK_RING / ICV_NO_FILL_RING) at the benchmark geometry: identical mask words, row counts and CSR arrays.
    ``ties``: rows equal to the reference and thresholds placed ON window values -- every such window is within one ulp
    of its threshold and goes through the float64 recomputation (list + k_thr_mask_ties here, in the kernel there)."""
    from infercnvpy_amd import _engine, _lib
    from infercnvpy_amd._plan import GenePlan

    torch = _engine._torch()
    lib = _lib.load()
    # "narrow": 338 windows -- rows of 1.3 KB, sixteen ring slots, two mask words per consumer wavefront
    genes = cases.GENES_PER_CHROM_20K if dtype != "narrow" else [1700, 1400, 520]
    dtype = np.float32 if dtype == "narrow" else dtype
    v = cases.synthetic_var(genes, seed_perm=3)
    X = cases.synthetic_expr(n, len(v["names"]), seed=n, dtype=dtype)
    ref_h = X.mean(axis=0).astype(dtype)
    if ties:
        X[5] = ref_h
        X[77] = ref_h
    dm = _engine.to_device_matrix(sp.csr_matrix(X) if fmt == "csr" else X)
    ref = torch.from_numpy(ref_h).cuda()
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    chunksize = 500
    try:
        res = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=dyn, chunksize=chunksize, apply=False)
        import ctypes as C

        geo = _lib.PackInfo()
        _lib.check(lib.icv_pack_geometry(plan.n_windows, C.byref(geo)))
        assert geo.mask_streamed and geo.fill_streamed and res.out.stride(0) % 4 == 0  # the streamed kernels do run here
        if ties:  # thresholds equal to |x_res| of a window of the chunk's first row, one and two ulps around it
            xr = res.out.cpu().numpy()
            thr = res.thr.cpu().numpy().copy()
            for c in range(thr.shape[0]):
                row = xr[c * chunksize]
                nz = np.flatnonzero(row)
                if nz.size:
                    a = np.abs(row[nz[c % nz.size]])
                    thr[c] = float(np.nextafter(a, np.float32([0, np.inf, 0][c % 3]), dtype=np.float32)) if c % 3 else float(a)
            res.thr.copy_(torch.from_numpy(thr))
        out = {}
        for tag, env in (("ring", {}), ("per_row", {"ICV_NO_MASK_RING": "1", "ICV_NO_FILL_RING": "1"})):
            for k in ("ICV_NO_MASK_RING", "ICV_NO_FILL_RING"):
                monkeypatch.delenv(k, raising=False)
            for k, val in env.items():
                monkeypatch.setenv(k, val)
            lib.icv_developer_knobs_reload()
            part = _engine.threshold_mask(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=chunksize)
            pk = _engine.threshold_csr(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=chunksize)
            nnz = pk.nnz()
            out[tag] = (part.mask.cpu().numpy(), part.counts.cpu().numpy(), pk.indptr.cpu().numpy(),
                        pk.indices[:nnz].cpu().numpy(), pk.data[:nnz].cpu().numpy())
        for a, b in zip(out["ring"], out["per_row"]):
            np.testing.assert_array_equal(a, b)
        assert out["ring"][1].sum() == out["ring"][2][-1]
        # twice in a row: the tie counters of the plan are back at zero after every call
        part2 = _engine.threshold_mask(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=chunksize)
        np.testing.assert_array_equal(part2.mask.cpu().numpy(), out["per_row"][0])
    finally:
        for k in ("ICV_NO_MASK_RING", "ICV_NO_FILL_RING"):
            monkeypatch.delenv(k, raising=False)
        lib.icv_developer_knobs_reload()
        plan.close()


@pytest.mark.parametrize("n", [0, 1, 4095, 4096, 4097, 12288, 1_000_003])
def test_row_offsets_equal_a_prefix_sum(n):
    """icv_row_offsets (k_row_block_sums + k_row_offsets over blocks of 4 096 rows) against torch.cumsum, at sizes on
    both sides of the block boundaries and with counts whose total passes 2^32."""
    from infercnvpy_amd import _engine, _lib

    torch = _engine._torch()
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n + 1)
    counts = torch.randint(0, 20_481, (max(n, 1),), device="cuda", generator=g, dtype=torch.int64)[:n]
    indptr = torch.full((n + 1,), -7, dtype=torch.int64, device="cuda")
    _lib.check(lib.icv_row_offsets(_engine._ptr(counts), n, _engine._ptr(indptr), _engine._stream_ptr(torch)))
    exp = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    if n:
        torch.cumsum(counts, 0, out=exp[1:])
    assert torch.equal(indptr, exp)
    if n >= 1_000_000:
        assert int(indptr[-1]) > 2**32

"""GPU parity tests (``-m gpu``): the HIP path, called through the C ABI, against

* the committed golden vectors captured from the reference (tests/golden/*.npz),
* the CPU oracle on seeded inputs at sizes it finishes in seconds,
* size-independent properties at the benchmark geometry.

Tolerance (BASELINE.json north_star): X_cnv within 1e-5 (float32 output of float64 arithmetic; we
additionally check 1e-6), gene/window indexing (chr_pos, window count, zero pattern of the
threshold step) bit-exact.
"""
import os

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import cases
from _golden import GoldenCase, case_names

pytestmark = pytest.mark.gpu

ATOL = 1e-5       # the stated tolerance
ATOL_TIGHT = 1e-6  # what float32 storage of |x| <= 3 actually allows (ulp(3)/2 = 1.2e-7)


def _knobs():
    """Have the library re-read its developer knobs (they are read once; this test has just changed the environment)."""
    from infercnvpy_amd import _lib

    _lib.load().icv_developer_knobs_reload()


def _adata(g):
    from infercnvpy_amd._compat import SimpleAnnData

    var = pd.DataFrame({"chromosome": g.chromosome, "start": g.start, "end": np.asarray(g.start) + 1000},
                       index=[f"g{i}" for i in range(len(g.start))])
    obs = pd.DataFrame(index=[f"c{i}" for i in range(g.X_dense.shape[0])])
    if g.obs is not None:
        obs["group"] = g.obs
    return SimpleAnnData(g.X, obs=obs, var=var)


_RUNNABLE = [n for n in case_names() if not n.startswith("genevals") and not n.startswith("mock4x10")]
EXACT_REFERENCE_CASES = [n for n in _RUNNABLE if "reference" in GoldenCase(n).kwargs]
MEAN_ON_GPU_CASES = [n for n in _RUNNABLE if n not in EXACT_REFERENCE_CASES]


@pytest.mark.parametrize("name", EXACT_REFERENCE_CASES)
def test_golden_explicit_reference(name):
    """Cases where the reference profile is an input: every threshold decision must match."""
    import infercnvpy_amd as cnv

    g = GoldenCase(name)
    chr_pos, res, per_gene = cnv.tl.infercnv(_adata(g), inplace=False, **g.api_kwargs())
    assert {k: int(v) for k, v in chr_pos.items()} == g.chr_pos
    assert list(chr_pos) == list(g.chr_pos)
    assert sp.issparse(res) and res.dtype == np.float64 and res.shape == g.out.shape
    got = res.toarray()
    np.testing.assert_array_equal(got == 0, g.out == 0)
    np.testing.assert_allclose(got, g.out, rtol=0, atol=ATOL_TIGHT)


SD_CASES = [n for n in _RUNNABLE if GoldenCase(n).fmt in ("csr", "csc") and GoldenCase(n).in_dtype == "float32"
            and GoldenCase(n).kwargs.get("window_size", 100) % 2 == 0
            and np.gcd(GoldenCase(n).kwargs.get("step", 10), GoldenCase(n).kwargs.get("window_size", 100) // 2) > 1
            and np.result_type(np.float32, np.asarray(GoldenCase(n).kwargs.get("reference", np.float32(0))).dtype) == np.float32]


@pytest.mark.parametrize("name", SD_CASES)
def test_sparse_float32_goldens_take_the_stored_entries_kernel(name, monkeypatch):
    """float32 CSR / CSC input in block form runs k_smooth_se (stored entries only: fixed-point block bins, windows
    from prefix sums) -- asserted through the plan's `last_kernel` record -- and its X_cnv equals that of the
    kernels that rebuild the row in LDS (developer knob ICV_NO_SD) up to the last float32 bit, zero pattern
    included.  (The golden comparison itself is test_golden_explicit_reference / test_golden_reference_mean_on_gpu.)"""
    import infercnvpy_amd as cnv
    from infercnvpy_amd import _lib

    g = GoldenCase(name)
    tm = {}
    _, res, _ = cnv.tl.infercnv(_adata(g), inplace=False, _timings=tm, **g.api_kwargs())
    assert tm["kernel"] == _lib.ICV_KERNEL_SD
    monkeypatch.setenv("ICV_NO_SD", "1")
    _knobs()
    tm2 = {}
    _, res2, _ = cnv.tl.infercnv(_adata(g), inplace=False, _timings=tm2, **g.api_kwargs())
    assert tm2["kernel"] in (_lib.ICV_KERNEL_WS_CSR, _lib.ICV_KERNEL_GENERIC)
    a, b = res.toarray(), res2.toarray()
    np.testing.assert_array_equal(a == 0, b == 0)
    np.testing.assert_allclose(a, b, rtol=0, atol=2.5e-7)


def _oracle_means(X, labels=None, cats=None):
    """The reference's own means (np.mean(X, axis=0) of the matrix AS STORED -- dense, CSR and CSC are three different
    evaluation orders, reference :385, :400), through the oracle's restatement of _get_reference."""
    from oracle import infercnv_oracle as O

    if cats is not None and isinstance(cats, str):
        cats = [cats]
    return np.asarray(O.reference_profile(X, labels, cats, None, X.shape[1]))


@pytest.mark.parametrize("name", MEAN_ON_GPU_CASES)
def test_golden_reference_mean_on_gpu(name):
    """Cases where the reference profile is computed from the matrix (reference :385, :400): the reference's DEFAULT
    call.  The GPU forms the means in numpy's / scipy's own evaluation order (sequential float32 chains per column for
    dense and CSR input, np.add.reduceat per column for CSC: icv_colchain / icv_colmean_csc), so every case -- the
    integer CSR matrix with two categories included, where a last-bit difference of a mean is amplified to whole units
    by the truncation of reference :428 -- must reproduce the captured reference output: identical zero pattern (no
    threshold flips), values to float32 storage."""
    import infercnvpy_amd as cnv
    from oracle import infercnv_oracle as O

    g = GoldenCase(name)
    chr_pos, res, _ = cnv.tl.infercnv(_adata(g), inplace=False, **g.api_kwargs())
    assert {k: int(v) for k, v in chr_pos.items()} == g.chr_pos
    got = res.toarray()
    # (a) the captured reference output
    np.testing.assert_array_equal(got == 0, g.out == 0)
    np.testing.assert_allclose(got, g.out, rtol=0, atol=ATOL_TIGHT)
    # (b) the oracle with ITS means of the matrix as stored
    kw = {k: v for k, v in g.array_kwargs().items() if k not in ("obs_col", "reference_cat")}
    ref = _oracle_means(g.X, g.obs, g.kwargs.get("reference_cat") if g.kwargs.get("reference_key") else None)
    _, exp, _, _ = O.infercnv(g.X, g.chromosome, g.start, reference=ref, **kw)
    exp = exp.toarray()
    np.testing.assert_array_equal(got == 0, exp == 0)
    np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_TIGHT)


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_reference_mean_goldens_bit_for_bit(fmt):
    """refmean_dense / refmean_csr: `_get_reference` of the reference itself on a float32 matrix (all cells; two
    categories), captured in the build container: the GPU's means are array_equal."""
    from infercnvpy_amd import _engine

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"refmean_{fmt}.npz"), allow_pickle=False)
    X, labels = z["X"], z["obs"]
    Xin = sp.csr_matrix(X) if fmt == "csr" else X
    dm = _engine.to_device_matrix(Xin)
    acc = _engine.column_chain(dm, None, None, X.shape[0])
    got = _engine.chain_mean(acc, X.shape[0], fmt == "csr").cpu().numpy()
    assert got.dtype == z["r_all"].dtype
    np.testing.assert_array_equal(got[None, :], z["r_all"])
    rows = []
    for c in ("normalA", "normalB"):
        sel = np.nonzero(labels == c)[0]
        acc = _engine.column_chain(dm, None, sel, len(sel))
        rows.append(_engine.chain_mean(acc, len(sel), fmt == "csr").cpu().numpy())
    np.testing.assert_array_equal(np.vstack(rows), z["r_cat"])


@pytest.mark.parametrize("name", [n for n in case_names() if n.startswith("genevals") or n.startswith("mock4x10")])
def test_golden_gene_values(name):
    """calculate_gene_values=True against the captured reference (incl. the reference's own 4 x 10 fixture,
    tests/test_tools.py:143-191): NaN pattern exact, values to float64 rounding."""
    import infercnvpy_amd as cnv

    g = GoldenCase(name)
    ad = _adata(g)
    chr_pos, res, per_gene = cnv.tl.infercnv(ad, inplace=False, **g.api_kwargs())
    assert {k: int(v) for k, v in chr_pos.items()} == g.chr_pos
    np.testing.assert_allclose(res.toarray(), g.out, rtol=0, atol=ATOL_TIGHT)
    assert per_gene.shape == g.per_gene.shape and per_gene.dtype == np.float64
    np.testing.assert_array_equal(np.isnan(per_gene), np.isnan(g.per_gene))
    np.testing.assert_array_equal(np.nan_to_num(per_gene) == 0, np.nan_to_num(g.per_gene) == 0)
    np.testing.assert_allclose(np.nan_to_num(per_gene), np.nan_to_num(g.per_gene), rtol=0, atol=1e-12)
    cnv.tl.infercnv(ad, **g.api_kwargs())
    assert "gene_values_cnv" in ad.layers and ad.layers["gene_values_cnv"].shape == g.per_gene.shape


def test_gene_values_against_oracle_medium():
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var([300, 120, 101, 60], extra=(("chrX", 10), (None, 3)))
    X = cases.synthetic_expr(40, len(v["names"]), seed=61)
    ref = X.mean(axis=0, dtype=np.float64).astype(np.float32)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    for fmt in (np.asarray, sp.csr_matrix):
        ad = SimpleAnnData(fmt(X), var=var)
        _, res, pg = cnv.tl.infercnv(ad, reference=ref, window_size=50, step=7, chunksize=16,
                                     calculate_gene_values=True, inplace=False)
        _, o_res, o_pg, _ = O.infercnv(fmt(X), v["chromosome"], v["start"], reference=ref, window_size=50, step=7,
                                       chunksize=16, calculate_gene_values=True)
        np.testing.assert_allclose(res.toarray(), o_res.toarray(), rtol=0, atol=ATOL_TIGHT)
        np.testing.assert_array_equal(np.isnan(pg), np.isnan(o_pg))
        np.testing.assert_array_equal(np.nan_to_num(pg) == 0, np.nan_to_num(o_pg) == 0)
        np.testing.assert_allclose(np.nan_to_num(pg), np.nan_to_num(o_pg), rtol=0, atol=1e-12)


@pytest.mark.parametrize("fmt,window,kernel", [("dense", 100, "X16"), ("dense", 250, "X16"), ("csr", 250, "SD"),
                                               ("csr", 100, "SD"), ("dense", 120, None)])
def test_gene_values_benchmark_geometry_one_smoothing_pass(fmt, window, kernel, monkeypatch):
    """calculate_gene_values at the 20 000-gene geometry (reference tl/_infercnv.py:247-298, :443-453): the float64
    windows come out of the SAME launch of the fast smoothing kernel that writes x_res (k_smooth_x16 / k_smooth_se;
    window 120: the generic kernel), the gene layer from ONE fused kernel.  Against the oracle: NaN and zero patterns
    exact, values to 1e-12; X_cnv unchanged by the flag; the fused kernel = the three round-1 kernels bit for bit;
    the device-resident call = the host call bit for bit."""
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd import _engine, _lib
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K, extra=(("chrX", 37), ("chrM", 3), (None, 4)))
    n_genes = len(v["names"]) - len(v["names"]) % 4
    for key in v:
        v[key] = v[key][:n_genes]
    n = 96
    X = cases.synthetic_expr(n, n_genes, seed=93)
    X[7, :] = 0.0  # a cell without stored entries
    ref = X.mean(axis=0, dtype=np.float64).astype(np.float32)
    Xin = sp.csr_matrix(X) if fmt == "csr" else X
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    kw = dict(reference=ref, window_size=window, step=10, chunksize=32)
    tm = {}
    pos, res, pg = cnv.tl.infercnv(SimpleAnnData(Xin, var=var), calculate_gene_values=True, inplace=False, _timings=tm, **kw)
    if kernel is not None:
        assert tm["kernel"] == getattr(_lib, "ICV_KERNEL_" + kernel)
    _, res_plain, none = cnv.tl.infercnv(SimpleAnnData(Xin, var=var), inplace=False, **kw)
    assert none is None
    for a, b in ((res.indptr, res_plain.indptr), (res.indices, res_plain.indices), (res.data, res_plain.data)):
        np.testing.assert_array_equal(a, b)
    _, o_res, o_pg, _ = O.infercnv(Xin, v["chromosome"], v["start"], calculate_gene_values=True, **kw)
    assert pg.shape == o_pg.shape == (n, n_genes) and pg.dtype == np.float64
    np.testing.assert_array_equal(np.isnan(pg), np.isnan(o_pg))
    np.testing.assert_array_equal(np.nan_to_num(pg) == 0, np.nan_to_num(o_pg) == 0)
    np.testing.assert_allclose(np.nan_to_num(pg), np.nan_to_num(o_pg), rtol=0, atol=1e-12)
    assert np.isnan(pg).any() and (np.nan_to_num(pg) != 0).any()
    # the round-1 kernels on the same windows: identical bits
    monkeypatch.setenv("ICV_NO_GENE_FUSED", "1")
    _lib.load().icv_developer_knobs_reload()
    _, _, pg_old = cnv.tl.infercnv(SimpleAnnData(Xin, var=var), calculate_gene_values=True, inplace=False, **kw)
    monkeypatch.delenv("ICV_NO_GENE_FUSED")
    _lib.load().icv_developer_knobs_reload()
    np.testing.assert_array_equal(pg_old.view(np.int64), pg.view(np.int64))
    # HBM-resident input: the same bits, everything stays on the device
    if fmt == "dense":
        Xd = torch.from_numpy(X).cuda()
    else:
        Xd = _engine.to_device_matrix(Xin)
    _, x_dev, pg_dev = cnv.tl.infercnv(SimpleAnnData(Xd, var=var), calculate_gene_values=True, inplace=False, **kw)
    assert pg_dev.is_cuda
    np.testing.assert_array_equal(pg_dev.cpu().numpy().view(np.int64), pg.view(np.int64))
    np.testing.assert_array_equal(x_dev.to_scipy().data, res.data)


def test_gene_values_old_entry_point_and_odd_strides():
    """icv_gene_values (smoothing + gene layer in one call) and icv_gene_values_from_windows with an odd row stride /
    an unaligned output (the 8-byte store path): equal to the aligned result."""
    import torch

    from infercnvpy_amd import _engine
    from infercnvpy_amd._plan import GenePlan

    v = cases.synthetic_var([700, 320, 150, 100, 61], extra=(("chrX", 40), (None, 3)))
    n_genes = len(v["names"])
    X = torch.from_numpy(cases.synthetic_expr(70, n_genes, seed=17)).cuda()
    dm = _engine.DeviceMatrix(dense=X)
    ref = X.mean(dim=0)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    try:
        res = _engine.run_hot_path(plan, dm, ref, chunksize=30, apply=False, windows=True)
        a = _engine.gene_values_from_windows(plan, res.windows, thr=res.thr, chunksize=30, n_vars=n_genes)
        b = _engine.gene_values(plan, dm, ref, thr=res.thr, chunksize=30)
        np.testing.assert_array_equal(a.cpu().numpy().view(np.int64), b.cpu().numpy().view(np.int64))
        wide = torch.full((70, n_genes + 3), 7.0, dtype=torch.float64, device="cuda")
        c = _engine.gene_values_from_windows(plan, res.windows, thr=res.thr, chunksize=30, out=wide[:, 1:1 + n_genes])
        np.testing.assert_array_equal(c.cpu().numpy().view(np.int64), a.cpu().numpy().view(np.int64))
        assert float(wide[:, 0].min()) == 7.0 and float(wide[:, -2:].min()) == 7.0  # nothing outside the view
        # no thresholds
        d = _engine.gene_values_from_windows(plan, res.windows, n_vars=n_genes)
        e = _engine.gene_values(plan, dm, ref)
        np.testing.assert_array_equal(d.cpu().numpy().view(np.int64), e.cpu().numpy().view(np.int64))
    finally:
        plan.close()


@pytest.mark.parametrize("window", [100, 250])
def test_position_ordered_columns_same_results(window):
    """SURVEY 7 "input column order != window order": ``adata.var`` in genome order (GTF order: chromosomes one after the
    other -- not in natural order, a masked chromosome in between --, positions ascending) against the oracle, and the same
    cells with the columns permuted give the same X_cnv bit for bit (the result does not depend on the column order)."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd import _lib
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v0 = cases.synthetic_var(cases.GENES_PER_CHROM_20K, extra=(("chrX", 37), ("chrM", 3)))
    order = [f"chr{i}" for i in (3, 1, 2)] + ["chrX"] + [f"chr{i}" for i in range(4, 23)] + ["chrM"]
    v, perm = cases.position_ordered(v0, chrom_order=order)
    n_genes = len(v["names"])
    n = 96
    X0 = cases.synthetic_expr(n, n_genes, seed=71)  # columns in v0's (random) order
    X0[5, :] = 0.0
    X0[11, 4321] = np.nan
    X = np.ascontiguousarray(X0[:, perm])           # the same cells, columns in genome order
    ref0 = np.nanmean(X0, axis=0).astype(np.float32)
    kw = dict(window_size=window, step=10, chunksize=40, inplace=False)
    var0 = pd.DataFrame({"chromosome": v0["chromosome"], "start": v0["start"], "end": v0["end"]}, index=v0["names"])
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    tm = {}
    pos, res, _ = cnv.tl.infercnv(SimpleAnnData(X, var=var), reference=ref0[perm], _timings=tm, **kw)
    assert tm["kernel"] == _lib.ICV_KERNEL_X16
    pos0, res0, _ = cnv.tl.infercnv(SimpleAnnData(X0, var=var0), reference=ref0, **kw)
    assert pos == pos0
    for a, b in ((res.indptr, res0.indptr), (res.indices, res0.indices), (res.data, res0.data)):
        np.testing.assert_array_equal(a, b)
    o_pos, o_res, _, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref0[perm], window_size=window, step=10,
                                    chunksize=40)
    assert {k: int(x) for k, x in pos.items()} == {k: int(x) for k, x in o_pos.items()}
    got, exp = res.toarray(), o_res.toarray()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_array_equal(np.nan_to_num(got) == 0, np.nan_to_num(exp) == 0)
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(exp), rtol=0, atol=ATOL_TIGHT)


def test_reference_fixture_through_public_api():
    """The reference's own 4 x 10 fixture (tests/conftest.py:61-108) through tl.infercnv, chunksize=2."""
    import infercnvpy_amd as cnv

    g = GoldenCase("mock4x10_chunks2")
    kw = g.api_kwargs()
    kw.pop("calculate_gene_values")
    ad = _adata(g)
    cnv.tl.infercnv(ad, **kw)
    expect = np.array([[1.00, 0, 0, 0, 0, 1.00], [-1.00, 0, 0, 0, 0, 0], [0, 1.25, 1.25, 0, 0, 0],
                       [0, 0, 0, 0.875, 0, 0]])
    np.testing.assert_array_equal(ad.obsm["X_cnv"].toarray(), expect)
    assert ad.uns["cnv"]["chr_pos"] == {"chr1": 0, "chr2": 3}


@pytest.mark.parametrize("fmt", ["dense", "csr"])
@pytest.mark.parametrize("window,step", [(100, 10), (250, 10), (101, 10)])
def test_against_oracle_benchmark_geometry(fmt, window, step):
    """1200 cells x 20 000 genes on chr1..22 (SURVEY §8(d) geometry), 3 std-chunks of 400 cells."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    n = 1200 if window != 101 else 300
    X = cases.synthetic_expr(n, 20000, seed=11)
    ref = X.mean(axis=0, dtype=np.float64).astype(np.float32)
    Xin = sp.csr_matrix(X) if fmt == "csr" else X
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    ad = SimpleAnnData(Xin, var=var)
    chr_pos, res, _ = cnv.tl.infercnv(ad, reference=ref, window_size=window, step=step, chunksize=400,
                                      inplace=False)
    o_pos, o_res, _, _ = O.infercnv(Xin, v["chromosome"], v["start"], reference=ref, window_size=window, step=step,
                                    chunksize=400, n_jobs=8)
    assert {k: int(x) for k, x in chr_pos.items()} == {k: int(x) for k, x in o_pos.items()}
    got, exp = res.toarray(), o_res.toarray()
    np.testing.assert_array_equal(got == 0, exp == 0)
    np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_TIGHT)


def test_float64_and_integer_inputs_against_oracle():
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var([900, 400, 260, 120], seed_start=3, seed_perm=4)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    labels = np.array(["a"] * 20 + ["b"] * 30 + ["t"] * 150)
    obs = pd.DataFrame({"group": labels}, index=[str(i) for i in range(200)])
    for X in (cases.synthetic_expr(200, 1680, seed=12, dtype=np.float64), cases.synthetic_counts(200, 1680, seed=13)):
        for kw in (dict(reference=X[:50].mean(axis=0)),
                   dict(reference=np.vstack([X[:20].mean(axis=0), X[20:50].mean(axis=0)]))):
            ad = SimpleAnnData(X, obs=obs, var=var)
            _, res, _ = cnv.tl.infercnv(ad, chunksize=64, inplace=False, **kw)
            _, o_res, _, _ = O.infercnv(X, v["chromosome"], v["start"], chunksize=64, **kw)
            got, exp = res.toarray(), o_res.toarray()
            np.testing.assert_array_equal(got == 0, exp == 0)
            np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_TIGHT)


def test_column_sums_and_reference_means():
    import torch

    from infercnvpy_amd import _engine

    rng = np.random.RandomState(0)
    X = rng.gamma(0.3, 1.0, size=(3001, 777)).astype(np.float32)
    groups = rng.randint(-1, 3, size=3001).astype(np.int32)
    for Xin in (X, sp.csr_matrix(X)):
        dm = _engine.to_device_matrix(Xin)
        s_all = _engine.column_sums(dm).cpu().numpy()[0]
        np.testing.assert_allclose(s_all, X.sum(axis=0, dtype=np.float64), rtol=1e-13)
        s_grp = _engine.column_sums(dm, groups, 3).cpu().numpy()
        for gi in range(3):
            np.testing.assert_allclose(s_grp[gi], X[groups == gi].sum(axis=0, dtype=np.float64), rtol=1e-13)
        # bitwise reproducible run to run, dense and CSR alike (no floating-point atomics whose order can vary)
        for _ in range(3):
            assert np.array_equal(_engine.column_sums(dm).cpu().numpy()[0], s_all)
            assert np.array_equal(_engine.column_sums(dm, groups, 3).cpu().numpy(), s_grp)
    torch.cuda.synchronize()


def test_cnv_score_known_answer_and_oracle():
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    x_cnv = np.array([[1, 1, 1, 2, 2, 1, 1, 1], [2, 2, 2, 1, 1, 2, 2, 2], [4, 4, 4, 2, 2, 3, 3, 3],
                      [2, 2, 2, 4, 4, 4, 4, 4]]).T
    obs = pd.DataFrame({"group": list("AAAAABBB")}, index=[f"c{i}" for i in range(8)])
    for wrap in (np.array, sp.csr_matrix, sp.csc_matrix):
        ad = SimpleAnnData(np.zeros((8, 3)), obs=obs.copy(), obsm={"X_cnv": wrap(x_cnv)})
        res = cnv.tl.cnv_score(ad, "group", inplace=False)
        assert res["A"] == pytest.approx(2.25, abs=1e-3) and res["B"] == pytest.approx(2.5, abs=1e-3)
        cnv.tl.cnv_score(ad, "group")
        np.testing.assert_allclose(ad.obs["cnv_score"].values, [2.25] * 5 + [2.5] * 3)
    rng = np.random.RandomState(1)
    big = rng.normal(size=(5000, 1802)).astype(np.float32)
    big[np.abs(big) < 1.2] = 0
    labels = rng.choice(["x", "y", "z"], size=5000)
    ad = SimpleAnnData(np.zeros((5000, 2)), obs=pd.DataFrame({"g": labels}), obsm={"X_cnv": sp.csr_matrix(big)})
    got = cnv.tl.cnv_score(ad, "g", inplace=False)
    exp = O.cnv_score(big.astype(np.float64), labels)
    for k in exp:
        assert got[k] == pytest.approx(exp[k], rel=1e-12)
    # a categorical groupby (what tl.leiden writes; an unused category, a missing label): the same numbers from its codes
    lab_c = pd.Series(labels).astype(pd.CategoricalDtype(["unused", "z", "y", "x"]))
    lab_c.iloc[7] = np.nan
    ad_c = SimpleAnnData(np.zeros((5000, 2)), obs=pd.DataFrame({"g": lab_c}), obsm={"X_cnv": sp.csr_matrix(big)})
    ad_s = SimpleAnnData(np.zeros((5000, 2)), obs=pd.DataFrame({"g": lab_c.astype(object)}), obsm={"X_cnv": sp.csr_matrix(big)})
    got_c, got_s = cnv.tl.cnv_score(ad_c, "g", inplace=False), cnv.tl.cnv_score(ad_s, "g", inplace=False)
    assert {k: v for k, v in got_c.items() if k == k} == {k: v for k, v in got_s.items() if k == k} and "unused" not in got_c
    cnv.tl.cnv_score(ad_c, "g")
    cnv.tl.cnv_score(ad_s, "g")
    np.testing.assert_array_equal(ad_c.obs["cnv_score"].values, ad_s.obs["cnv_score"].values)
    assert np.isnan(ad_c.obs["cnv_score"].values[7])
    with pytest.raises(ValueError):
        cnv.tl.cnv_score(ad)
    with pytest.warns(FutureWarning):
        cnv.tl.cnv_score(ad, obs_key="g", inplace=False)


def test_config1_like_step1_direct_form():
    """BASELINE config 1 stand-in (the oligodendroglioma file is not shipped): 183 cells x 11 000 genes,
    window 100, step 1 (the notebook's setting) -> ~9 000 windows, direct-form generic kernel; dense, CSR
    and CSC inputs give the same result (reference tests/conftest.py:17-24 parametrisation)."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    genes = [1290, 730, 610, 430, 500, 590, 520, 390, 445, 420, 730, 580, 190, 375, 345, 495, 665, 160, 825, 315,
             140, 255]
    v = cases.synthetic_var(genes, seed_start=5, seed_perm=6, extra=(("chrX", 300), ("chrM", 13), (None, 40)))
    X = cases.synthetic_expr(183, len(v["names"]), seed=81)
    labels = np.array(["Microglia"] * 30 + ["Oligo"] * 25 + ["tumor"] * 128)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    obs = pd.DataFrame({"cell_type": labels}, index=[f"c{i}" for i in range(183)])
    for wrap in (np.asarray, sp.csr_matrix, sp.csc_matrix):
        # the means are those of the matrix as stored: numpy (dense), scipy CSR and scipy CSC add in three orders
        ref = _oracle_means(wrap(X), labels, ["Microglia", "Oligo"])
        o_pos, o_res, _, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref, step=1)
        exp = o_res.toarray()
        assert exp.shape[1] == sum(g - 99 for g in genes)
        ad = SimpleAnnData(wrap(X), obs=obs.copy(), var=var)
        cnv.tl.infercnv(ad, reference_key="cell_type", reference_cat=["Microglia", "Oligo"], step=1)
        got = ad.obsm["X_cnv"].toarray()
        assert {k: int(x) for k, x in ad.uns["cnv"]["chr_pos"].items()} == {k: int(x) for k, x in o_pos.items()}
        np.testing.assert_array_equal(got == 0, exp == 0)
        np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_TIGHT)


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_large_gene_count_generic_path(fmt):
    """30 000 genes do not fit the register-prefetch kernels (G <= 20 480): the generic kernel takes
    over with one workgroup per CU (120 KB LDS row)."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    genes = [int(g * 1.5) for g in cases.GENES_PER_CHROM_20K]
    v = cases.synthetic_var(genes, seed_start=7, seed_perm=8)
    X = cases.synthetic_expr(150, sum(genes), seed=82)
    ref = X.mean(axis=0, dtype=np.float64).astype(np.float32)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    Xin = sp.csr_matrix(X) if fmt == "csr" else X
    _, res, _ = cnv.tl.infercnv(SimpleAnnData(Xin, var=var), reference=ref, chunksize=64, inplace=False)
    _, o_res, _, _ = O.infercnv(Xin, v["chromosome"], v["start"], reference=ref, chunksize=64, n_jobs=8)
    np.testing.assert_array_equal(res.toarray() == 0, o_res.toarray() == 0)
    np.testing.assert_allclose(res.toarray(), o_res.toarray(), rtol=0, atol=ATOL_TIGHT)


def test_edge_shapes():
    """Empty matrix, a single cell, fewer cells than a chunk, all-zero matrix."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var([300, 120, 101, 60])
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    n_genes = len(v["names"])
    ref = np.linspace(0.0, 1.0, n_genes).astype(np.float32)
    for n in (0, 1, 3):
        for fmt in (np.asarray, sp.csr_matrix):
            X = cases.synthetic_expr(max(n, 1), n_genes, seed=71)[:n]
            chr_pos, res, _ = cnv.tl.infercnv(SimpleAnnData(fmt(X), var=var), reference=ref, inplace=False)
            assert res.shape == (n, 26) and list(chr_pos) == ["chr1", "chr2", "chr3", "chr4"]
            if n:
                _, o_res, _, _ = O.infercnv(fmt(X), v["chromosome"], v["start"], reference=ref)
                np.testing.assert_array_equal(res.toarray() == 0, o_res.toarray() == 0)
                np.testing.assert_allclose(res.toarray(), o_res.toarray(), rtol=0, atol=ATOL_TIGHT)
    Z = np.zeros((5, n_genes), dtype=np.float32)
    _, res, _ = cnv.tl.infercnv(SimpleAnnData(Z, var=var), reference=np.zeros(n_genes, np.float32), inplace=False)
    assert res.nnz == 0 and res.shape == (5, 26)


def test_error_behaviour_matches_reference():
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    g = GoldenCase("m_dense_f32_r1")
    ad = _adata(g)
    ad.var = ad.var.drop(columns=["start"])
    with pytest.raises(ValueError):
        cnv.tl.infercnv(ad)
    ad = _adata(g)
    with pytest.raises(ValueError):  # wrong reference length (reference :405-406)
        cnv.tl.infercnv(ad, reference=np.ones(3))
    ad = _adata(GoldenCase("m_dense_f32_r2"))
    with pytest.raises(ValueError):  # unknown category (reference :393-398)
        cnv.tl.infercnv(ad, reference_key="group", reference_cat=["normalA", "nope"])
    ad = _adata(g)
    ad.var.index = ["dup"] * len(ad.var)
    with pytest.raises(ValueError):
        cnv.tl.infercnv(ad)
    # layer= gives the same result as X= (reference tests/test_tools.py:221-239)
    ad = _adata(g)
    ad.layers["counts"] = ad.X.copy()
    cnv.tl.infercnv(ad, reference=g.kwargs["reference"], key_added="a")
    cnv.tl.infercnv(ad, reference=g.kwargs["reference"], layer="counts", key_added="b")
    np.testing.assert_array_equal(ad.obsm["X_a"].toarray(), ad.obsm["X_b"].toarray())


def test_fast_and_generic_kernels_are_bit_identical(monkeypatch):
    """k_smooth_x16 (16 wavefronts per cell, the default for the window 100 / step 10 geometry), k_smooth_ws
    (two 8-wavefront workgroups per CU, pipelined histogram median) and k_smooth (generic) share one float64
    evaluation order for windows and median: outputs and medians must agree bit for bit.  The per-cell
    moments are reduced in kernel-specific (fixed) orders, so they and the thresholds agree to float64
    rounding."""
    import torch

    from infercnvpy_amd import _engine
    from infercnvpy_amd._plan import GenePlan

    def run(plan, dm, ref, env, stats=True):
        for k in ("ICV_FORCE_GENERIC", "ICV_NO_X16"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, "1")
        _knobs()
        res = _engine.run_hot_path(plan, dm, ref, chunksize=300, cell_stats=stats)
        torch.cuda.synchronize()
        for k in env:
            monkeypatch.delenv(k)
        _knobs()
        return res

    for genes, window, step in ((cases.GENES_PER_CHROM_20K, 100, 10), (cases.GENES_PER_CHROM_20K, 250, 10),
                                ([230, 110, 101, 100, 99, 57, 140], 100, 10), ([600, 260, 251, 250, 249], 20, 4)):
        v = cases.synthetic_var(genes, extra=(("chrX", 31), (None, 3)))
        n_genes = len(v["names"])
        n_genes4 = n_genes - n_genes % 4
        for key in ("chromosome", "start"):
            v[key] = v[key][:n_genes4]
        plan = GenePlan(v["chromosome"], v["start"], window_size=window, step=step)
        X = torch.from_numpy(cases.synthetic_expr(700, n_genes4, seed=21)).cuda()
        ref = X[7:].mean(dim=0)
        X[5] = ref  # centred row is all zero: every window equal (> 64 ties at the median)
        X[6, 17] = float("nan")
        dm = _engine.DeviceMatrix(dense=X)
        gen = run(plan, dm, ref, ["ICV_FORCE_GENERIC"])
        dm_csr = _engine.to_device_matrix(sp.csr_matrix(X.cpu().numpy()))  # prepared-entry CSR fast path
        # default (k_smooth_x16 where the geometry admits it, else k_smooth_ws), k_smooth_ws forced,
        # prepared-entry CSR, generic CSR
        # float32 CSR input: k_smooth_se adds the stored entries' differences to the zero row into block bins and
        # reads the windows off prefix sums -- another float64 evaluation order, equal to the canonical one to
        # ~1e-12 (float32 x_res: last-bit differences in a few entries per 100 000)
        # (every block-form float32 CSR geometry runs k_smooth_se; ICV_NO_SD: the prepared-entry kernel k_smooth_ws<CSR>
        # where the geometry admits it, else the generic kernel -- both in the canonical order)
        from infercnvpy_amd import _lib

        for env, mat, exact in (([], dm, True), (["ICV_NO_X16"], dm, True), ([], dm_csr, False),
                                (["ICV_NO_SD"], dm_csr, True), (["ICV_FORCE_GENERIC"], dm_csr, True)):
            def same(a, b, what):
                a, b = torch.nan_to_num(a, nan=123.0), torch.nan_to_num(b, nan=123.0)
                if exact:
                    assert torch.equal(a, b), (env, what)
                else:
                    torch.testing.assert_close(a.double(), b.double(), rtol=0, atol=2.5e-7 if what == "out" else 1e-11)
                    if what == "out":
                        assert (a != b).double().mean().item() < 1e-3, env

            tol = 1e-12 if exact else 1e-9
            fast = run(plan, mat, ref, env)
            if mat is dm_csr:
                assert plan.last_kernel() == (_lib.ICV_KERNEL_SD if not env else
                                              (_lib.ICV_KERNEL_GENERIC if env == ["ICV_FORCE_GENERIC"] else plan.last_kernel()))
            same(fast.out, gen.out, "out")
            same(fast.cell_median, gen.cell_median, "median")
            for a, b in ((fast.cell_stats, gen.cell_stats), (fast.thr, gen.thr)):
                torch.testing.assert_close(a, b, rtol=tol, atol=tol, equal_nan=True)
            assert torch.isnan(fast.out[6]).all() and not torch.isnan(fast.out[:6]).any()
            assert (fast.out[5] == 0).all()
            # without per-cell moments the thresholds come from per-chunk partial sums (formed inside
            # k_smooth_x16 where it runs): same thresholds to rounding, same output
            lean = run(plan, mat, ref, env, stats=False)
            assert lean.cell_stats is None
            torch.testing.assert_close(lean.thr, gen.thr, rtol=tol, atol=tol, equal_nan=True)
            d = torch.nan_to_num(lean.out, nan=123.0) != torch.nan_to_num(gen.out, nan=123.0)
            if exact:
                assert not d.any(), env
            else:  # thresholded: an entry within ~1e-12 of the threshold may fall on the other side
                assert d.double().mean().item() < 1e-3, env


def test_csr_long_windows_do_not_depend_on_entry_order():
    """k_smooth_se accumulates in fixed point: any order of a row's stored entries gives the same bits, and so do
    repeated runs.  Geometries: the benchmark's (window 250 / step 10; window 100 / step 10) and one with other block
    sizes, masked columns and a chromosome shorter than the window (flat window)."""
    import torch

    from infercnvpy_amd import _engine
    from infercnvpy_amd._plan import GenePlan

    for genes, window, step, extra in ((cases.GENES_PER_CHROM_20K, 250, 10, (("chrX", 31), (None, 3))),
                                       (cases.GENES_PER_CHROM_20K, 100, 10, (("chrX", 30),)),
                                       ([1500, 700, 333, 90], 120, 4, ((None, 5),))):
        v = cases.synthetic_var(genes, extra=extra)
        n_genes = len(v["names"]) - len(v["names"]) % 4
        for key in ("chromosome", "start"):
            v[key] = v[key][:n_genes]
        plan = GenePlan(v["chromosome"], v["start"], window_size=window, step=step)
        Xh = cases.synthetic_expr(300, n_genes, seed=5)
        Xh[Xh < np.quantile(Xh, 0.9)] = 0  # ~10 % stored
        ref = torch.from_numpy(Xh[100:].mean(axis=0)).cuda()
        csr = sp.csr_matrix(Xh)
        rng = np.random.RandomState(0)
        indices, data = csr.indices.copy(), csr.data.copy()
        for r in range(csr.shape[0]):  # shuffle the entries inside every row
            a, b = csr.indptr[r], csr.indptr[r + 1]
            perm = rng.permutation(b - a)
            indices[a:b], data[a:b] = indices[a:b][perm], data[a:b][perm]

        def run(ix, dv):
            dm = _engine.DeviceMatrix(indptr=torch.from_numpy(csr.indptr.astype(np.int64)).cuda(),
                                      indices=torch.from_numpy(ix.astype(np.int32)).cuda(),
                                      data=torch.from_numpy(dv.astype(np.float32)).cuda(), shape=csr.shape,
                                      validate=False)  # (unsorted on purpose: the public constructor refuses it)
            res = _engine.run_hot_path(plan, dm, ref, chunksize=100, cell_stats=True)
            torch.cuda.synchronize()
            return res

        base, again, shuffled = run(csr.indices, csr.data), run(csr.indices, csr.data), run(indices, data)
        from infercnvpy_amd import _lib

        assert plan.last_kernel() == _lib.ICV_KERNEL_SD
        for other in (again, shuffled):
            assert torch.equal(base.out, other.out)
            assert torch.equal(base.cell_median, other.cell_median)
            assert torch.equal(base.cell_stats, other.cell_stats)
        # and the values are those of the dense path
        dense = _engine.run_hot_path(plan, _engine.DeviceMatrix(dense=torch.from_numpy(Xh).cuda()), ref, chunksize=100,
                                     cell_stats=True)
        torch.testing.assert_close(base.out.double(), dense.out.double(), rtol=0, atol=2.5e-7)
        torch.testing.assert_close(base.cell_median, dense.cell_median, rtol=0, atol=1e-11)


def test_csr_rows_without_entries_and_ragged_rows():
    """The stored-entries kernel on ragged input: rows without any stored entry (their windows are the zero row's),
    rows with one entry, rows with more entries than the prefetch covers (> 2048), explicit zeros among the stored
    values, at the benchmark geometry (windows that cross into the next wavefront's blocks) -- against the oracle."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd import _lib
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K, extra=(("chrX", 30), (None, 2)))
    n_genes = len(v["names"])
    rng = np.random.RandomState(12)
    X = np.zeros((41, n_genes), dtype=np.float32)
    for r in range(41):
        nnz = [0, 1, 2, 7, 300, 1400, 2047, 2048, 2049, 5000, n_genes][r % 11]
        cols = rng.choice(n_genes, size=nnz, replace=False)
        X[r, cols] = rng.gamma(0.5, 1.0, size=nnz).astype(np.float32) + 0.01
    ref = (rng.gamma(0.3, 0.2, size=n_genes)).astype(np.float32)
    Xs = sp.csr_matrix(X)
    # explicit zeros stored in the matrix: a stored 0 is still the value 0 (x - ref), like an implicit one
    Xs.data[::17] = 0.0
    Xd = Xs.toarray()
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    for window, step in ((100, 10), (250, 10)):
        tm = {}
        pos, res, _ = cnv.tl.infercnv(SimpleAnnData(Xs.copy(), var=var), reference=ref, window_size=window, step=step,
                                      chunksize=16, inplace=False, _timings=tm)
        assert tm["kernel"] == _lib.ICV_KERNEL_SD
        e_pos, e_res, _, _ = O.infercnv(Xd, v["chromosome"], v["start"], reference=ref, window_size=window, step=step,
                                        chunksize=16)
        assert {k: int(p) for k, p in pos.items()} == {k: int(p) for k, p in e_pos.items()}
        got, exp = res.toarray(), e_res.toarray()
        np.testing.assert_array_equal(got == 0, exp == 0)
        np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_TIGHT)


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_threshold_ties_are_decided_in_float64(fmt):
    """Force the rare tie path of k_apply_thr: choose thr = (double)float32(|x|) of existing entries, so
    the stored float32 cannot decide and the window is recomputed from the input in float64.  The
    decision must be the oracle's |x64| < thr."""
    import ctypes as C

    import torch

    from infercnvpy_amd import _engine, _lib
    from infercnvpy_amd._plan import GenePlan
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var([900, 400, 260, 120], seed_start=3, seed_perm=4)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    Xh = cases.synthetic_expr(64, 1680, seed=51)
    refh = Xh.mean(axis=0, dtype=np.float64).astype(np.float32)
    dm = _engine.to_device_matrix(sp.csr_matrix(Xh) if fmt == "csr" else Xh)
    ref = torch.from_numpy(refh).cuda()
    raw = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=None)
    _, x64, _, _ = O.infercnv_chunk(Xh, v["chromosome"], v["start"], refh[None, :], 3, 100, 10, None)
    out0 = raw.out.clone()
    lib = _lib.load()
    checked = 0
    for row in range(0, 64, 7):
        for col in (3, 40, 101):
            thr_val = float(abs(out0[row, col].item()))  # exactly a float32 value
            if thr_val == 0.0:
                continue
            out = out0.clone()
            thr = torch.full((64,), thr_val, dtype=torch.float64, device="cuda")
            m = dm.c_struct()
            _lib.check(lib.icv_apply_threshold(
                plan.handle, C.byref(m), _engine._ptr(ref), None, 3.0, 0, _engine._ptr(out), out.stride(0),
                _engine._ptr(raw.cell_median), _engine._ptr(thr), 1, 0, _engine._stream_ptr(torch)))
            torch.cuda.synchronize()
            got_zero = out[row, col].item() == 0.0
            assert got_zero == (abs(x64[row, col]) < thr_val), (row, col)
            # every entry of the row that does not tie follows the plain float32 comparison
            free = out0[row].abs() != thr_val
            assert torch.equal((out[row] == 0)[free], ((out0[row].abs() < thr_val) | (out0[row] == 0))[free])
            checked += 1
    assert checked >= 20


def test_unaligned_shards_match_single_run():
    """Two row shards cut in the middle of a std-chunk (what ``dist.run_shard`` does per rank, here
    sequentially on one GPU): chunk moments summed across shards give the same thresholds and the
    same thresholded rows as one run over all rows."""
    import ctypes as C

    import torch

    from infercnvpy_amd import _engine, _lib, dist as icd
    from infercnvpy_amd._plan import GenePlan

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    X = torch.from_numpy(cases.synthetic_expr(1100, 20000, seed=41)).cuda()
    ref = X.mean(dim=0)
    cs = 400
    whole = _engine.run_hot_path(plan, _engine.DeviceMatrix(dense=X), ref, chunksize=cs)
    cut = 537
    parts, moments = [], []
    for r0, r1 in ((0, cut), (cut, 1100)):
        dm = _engine.DeviceMatrix(dense=X[r0:r1].contiguous())
        res = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=None, cell_stats=True)
        parts.append((r0, r1, dm, res))
        moments.append(icd.chunk_moments(res.cell_stats, r0, cs, 3))
    thr_all = icd.thresholds_from_moments(moments[0] + moments[1], plan.n_windows, 1.5)
    np.testing.assert_allclose(thr_all.cpu().numpy(), whole.thr.cpu().numpy(), rtol=1e-13)
    lib = _lib.load()
    for r0, r1, dm, res in parts:
        k0 = r0 // cs
        thr = thr_all[k0:(r1 - 1) // cs + 1].contiguous()
        m = dm.c_struct()
        _lib.check(lib.icv_apply_threshold(
            plan.handle, C.byref(m), _engine._ptr(ref), None, 3.0, 0, _engine._ptr(res.out), res.out.stride(0),
            _engine._ptr(res.cell_median), _engine._ptr(thr), cs, r0 % cs, _engine._stream_ptr(torch)))
        torch.cuda.synchronize()
        assert torch.equal(res.out, whole.out[r0:r1])
    # run_shard with a chunk-aligned shard and no process group is the plain single-GPU path
    dm = _engine.DeviceMatrix(dense=X[400:800].contiguous())
    res = icd.run_shard(plan, dm, ref, global_row0=400, n_obs_global=1100, chunksize=cs)
    assert torch.equal(res.out, whole.out[400:800])


# ---- properties at the benchmark size ---------------------------------------------------------
def _bench_inputs(n_cells, seed=2):
    import torch

    from infercnvpy_amd._plan import GenePlan

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    X = torch._standard_gamma(torch.full((n_cells, 20000), 0.3, device="cuda"), generator=gen) \
        if hasattr(torch, "_standard_gamma") else torch.rand((n_cells, 20000), device="cuda", generator=gen)
    X = torch.where(X < 0.5, torch.zeros_like(X), X).float().contiguous()
    return v, plan, X


def test_full_size_properties():
    """100 000 x 20 000 float32 (8 GB, BASELINE config 2): determinism, chunk structure,
    median-centring, threshold consistency, and sampled rows against the oracle."""
    import torch

    from infercnvpy_amd import _engine
    from oracle import infercnv_oracle as O

    n = 100_000
    v, plan, X = _bench_inputs(n)
    dm = _engine.DeviceMatrix(dense=X)
    sums = _engine.column_sums(dm)
    ref = (sums[0] / n).float()
    np.testing.assert_allclose(ref.cpu().numpy(), X.double().mean(dim=0).cpu().numpy(), rtol=2e-7)

    raw = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=None)
    thr_run = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=1.5, chunksize=5000)
    thr_run2 = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=1.5, chunksize=5000)
    torch.cuda.synchronize()
    # (1) bitwise deterministic
    assert torch.equal(thr_run.out, thr_run2.out) and torch.equal(thr_run.thr, thr_run2.thr)
    # (2) median-centred: per-row median of x_res is 0 to float32 rounding
    srt = raw.out[:2000].double().sort(dim=1).values
    med = (0.5 * (srt[:, 900] + srt[:, 901])).abs().max().item()  # W = 1802: mean of the two middle values
    assert med < 1e-6
    # (3) thresholding == zeroing |x| < thr[chunk] of the un-thresholded result, chunk by chunk
    thr = thr_run.thr.cpu().numpy()
    assert thr.shape == (20,)
    for k in (0, 7, 19):
        a = raw.out[k * 5000:(k + 1) * 5000]
        b = thr_run.out[k * 5000:(k + 1) * 5000]
        std = a.double().std(unbiased=False).item()
        assert thr[k] == pytest.approx(1.5 * std, rel=1e-6)
        keep = a.abs().double() >= thr[k] * (1 + 1e-6)
        drop = a.abs().double() <= thr[k] * (1 - 1e-6)
        assert torch.equal(b[keep], a[keep]) and (b[drop] == 0).all()
    # (4) rows are independent: a row computed alone equals the row computed in the batch
    rows = [0, 1, 4999, 5000, 54321, n - 1]
    sub = _engine.DeviceMatrix(dense=X[rows].contiguous())
    alone = _engine.run_hot_path(plan, sub, ref, dynamic_threshold=None)
    assert torch.equal(alone.out, raw.out[rows])
    # (5) sampled rows against the oracle (threshold disabled: the oracle cannot afford a chunk)
    xs = X[rows].cpu().numpy()
    _, o_res, _, _ = O.infercnv(xs, v["chromosome"], v["start"], reference=ref.cpu().numpy(), dynamic_threshold=None)
    np.testing.assert_allclose(alone.out.cpu().numpy(), o_res.toarray(), rtol=0, atol=ATOL_TIGHT)
    # (6) shifting matrix and reference by the same constant changes nothing beyond float32 rounding
    shifted = _engine.run_hot_path(plan, _engine.DeviceMatrix(dense=(X[rows] + 0.25).contiguous()), ref + 0.25,
                                   dynamic_threshold=None)
    np.testing.assert_allclose(shifted.out.cpu().numpy(), alone.out.cpu().numpy(), rtol=0, atol=2e-6)


def test_ith_scores_known_answers_golden_and_oracle():
    """ithgex / ithcna (reference tests/test_scores.py:6-15) on the MFMA correlation path.

    Tolerance 1e-5 absolute on the score: the Gram matrix is accumulated in float32 (rows normalised
    with float64 statistics), numpy's corrcoef in float64.
    """
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    X = np.array([[1, 1, 1, 1, 1, 1, 2, 3], [2, 2, 2, 2, 2, 2, 8, 0], [3, 3, 3, 3, 3, 10, 3, 7]]).T
    x_cnv = np.array([[1, 1, 1, 2, 2, 1, 1, 1], [2, 2, 2, 1, 1, 2, 2, 2], [4, 4, 4, 2, 2, 3, 3, 3],
                      [2, 2, 2, 4, 4, 4, 4, 4]]).T
    obs = pd.DataFrame({"group": list("AAAAABBB")}, index=[f"c{i}" for i in range(8)])
    for wrap in (np.array, sp.csr_matrix, sp.csc_matrix):
        ad = SimpleAnnData(wrap(X), obs=obs.copy(), obsm={"X_cnv": wrap(x_cnv)})
        res = cnv.tl.ithgex(ad, "group", inplace=False)
        assert res["A"] == 0 and res["B"] == pytest.approx(1.2628, abs=1e-3)
        assert res["B"] == pytest.approx(O.ith_score(X, list("AAAAABBB"))["B"], abs=1e-5)
        res = cnv.tl.ithcna(ad, "group", inplace=False)
        assert res["A"] == pytest.approx(1.053, abs=1e-3) and res["B"] == 0
        cnv.tl.ithcna(ad, "group")
        cnv.tl.ithgex(ad, "group", key_added="gex")
        assert ad.obs["ithcna"].values[0] == pytest.approx(1.053, abs=1e-3) and ad.obs["gex"].values[-1] == pytest.approx(1.2628, abs=1e-3)
    with pytest.raises(ValueError):
        cnv.tl.ithgex(ad, "group", use_raw=True, layer="counts")

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ith_scores.npz"), allow_pickle=False)
    Xg, labels = z["X"], z["labels"]
    ad = SimpleAnnData(Xg, obs=pd.DataFrame({"group": labels}),
                       obsm={"X_cnv": sp.csr_matrix(np.where(np.abs(Xg) > 0.8, Xg, 0).astype(np.float64))})
    gex = cnv.tl.ithgex(ad, "group", inplace=False)
    cna = cnv.tl.ithcna(ad, "group", inplace=False)
    assert sorted(gex) == list(z["keys"]) == sorted(cna)
    for i, g in enumerate(z["keys"]):
        np.testing.assert_allclose(gex[g], z["gex"][i], rtol=0, atol=1e-5)
        np.testing.assert_allclose(cna[g], z["cna"][i], rtol=0, atol=1e-5, equal_nan=True)
    with pytest.raises(KeyError):  # the reference fails the same way on a single-cell group (:148)
        cnv.tl.ithgex(ad, "group")

    # ragged sizes (n not a multiple of the 128 tile, k not a multiple of 16) against numpy
    rng = np.random.RandomState(5)
    for n, k in ((2, 3), (129, 17), (700, 1802), (2500, 4000)):
        Xr = (rng.standard_normal((n, k)) + rng.standard_normal((1, k))).astype(np.float32)
        got = cnv.tl.ithcna(SimpleAnnData(np.zeros((n, 1)), obs=pd.DataFrame({"g": ["a"] * n}), obsm={"X_cnv": Xr}),
                            "g", inplace=False)["a"]
        assert got == pytest.approx(O.ith_score(Xr, ["a"] * n)["a"], abs=1e-5)


# --------------------------------------------------------------------------- #
# BASELINE config 5: pairwise Euclidean (fp32 MFMA) + Ward linkage; oracle = scipy (parity unpinned
# against the reference, which has no call site for cell-level clustering)
# --------------------------------------------------------------------------- #
def _blobs(n, d, k, seed, spread=4.0):
    rng = np.random.RandomState(seed)
    centres = rng.standard_normal((k, d)) * spread
    return (centres[rng.randint(0, k, n)] + rng.standard_normal((n, d))).astype(np.float32)


def _cluster_hashes(Z):
    n = Z.shape[0] + 1
    key = np.random.RandomState(0).randint(1, 2 ** 62, size=n, dtype=np.int64).astype(np.uint64)
    h = np.concatenate([key, np.zeros(n - 1, dtype=np.uint64)])
    for q in range(n - 1):
        h[n + q] = h[int(Z[q, 0])] + h[int(Z[q, 1])]  # wraps mod 2^64
    return set(h[n:].tolist())


@pytest.mark.parametrize("n,d", [(2, 3), (3, 1), (129, 17), (700, 64), (1500, 1802)])
def test_pairwise_sqeuclidean_matches_float64(n, d):
    import torch
    from infercnvpy_amd import _engine

    X = _blobs(n, d, 5, seed=n) + 3.0  # off-centre on purpose: the kernel centres the columns
    got = _engine.pairwise_sqeuclidean(torch.from_numpy(X).cuda()).cpu().numpy()
    x64 = X.astype(np.float64)
    exp = ((x64[:, None, :] - x64[None, :, :]) ** 2).sum(-1) if n <= 700 else None
    if exp is None:
        g = x64 @ x64.T
        exp = np.maximum(np.diag(g)[:, None] + np.diag(g)[None, :] - 2 * g, 0)
    np.testing.assert_array_equal(got, got.T)  # bit-exact symmetry (the Ward rounds rely on it)
    assert (np.diag(got) == 0).all()
    # tolerance: 1e-5 relative to the scale of the squared norms entering the difference
    scale = exp.max()
    np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("n,d,k", [(2, 4, 1), (3, 2, 1), (10, 3, 2), (257, 20, 6), (1000, 50, 8), (3000, 300, 12)])
def test_ward_linkage_matches_scipy(n, d, k):
    import infercnvpy_amd as cnv
    from oracle import infercnv_oracle as O
    from scipy.cluster.hierarchy import fcluster, is_valid_linkage

    X = _blobs(n, d, k, seed=100 + n)
    Z, rounds = cnv.tl.ward_linkage(X, return_rounds=True)
    Zs = O.ward_linkage(X)
    assert Z.shape == Zs.shape == (n - 1, 4) and is_valid_linkage(Z)
    assert rounds <= n - 1
    # heights: float32 distance matrix vs scipy's float64 -> 1e-4 relative (documented in DESIGN.md)
    np.testing.assert_allclose(Z[:, 2], Zs[:, 2], rtol=1e-4)
    np.testing.assert_array_equal(np.sort(Z[:, 3]), np.sort(Zs[:, 3]))
    # same tree: identical flat clusterings at several cuts (merge ids can only permute between
    # merges whose heights agree to rounding)
    for kk in {2, 3, min(k, n), min(2 * k, n), min(50, n)}:
        a = fcluster(Z, kk, "maxclust")
        b = fcluster(Zs, kk, "maxclust")
        assert len(set(zip(a, b))) == len(set(a)) == len(set(b))
    # same tree topology: the leaf set of every merged cluster (hashed) appears in scipy's tree too; cluster ids
    # themselves may permute between merges whose heights agree to rounding
    mine, ref = _cluster_hashes(Z), _cluster_hashes(Zs)
    assert len(mine & ref) >= (0.999 if n > 500 else 1.0) * (n - 1)


def test_ward_linkage_duplicates_sparse_and_errors():
    import infercnvpy_amd as cnv
    from oracle import infercnv_oracle as O

    X = _blobs(60, 8, 3, seed=4)
    X[10:20] = X[10]  # duplicate cells: zero distances, ties
    Z = cnv.tl.ward_linkage(sp.csr_matrix(X))
    Zs = O.ward_linkage(X)
    np.testing.assert_allclose(Z[:, 2], Zs[:, 2], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(np.sort(Z[:, 3]), np.sort(Zs[:, 3]))
    bad = X.copy()
    bad[3, 2] = np.nan
    with pytest.raises(ValueError):
        cnv.tl.ward_linkage(bad)
    with pytest.raises(ValueError):
        cnv.tl.ward_linkage(X[:1])


def test_cell_linkage_and_heatmap_dendrogram():
    import matplotlib

    matplotlib.use("Agg")
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from scipy.cluster.hierarchy import leaves_list

    X = _blobs(300, 40, 4, seed=9)
    X[np.abs(X) < 0.5] = 0
    ad = SimpleAnnData(np.zeros((300, 2)), obs=pd.DataFrame({"g": np.repeat(["a", "b", "c"], 100)}),
                       obsm={"X_cnv": sp.csr_matrix(X.astype(np.float64))},
                       uns={"cnv": {"chr_pos": {"chr1": 0, "chr2": 25}}})
    cnv.tl.cell_linkage(ad)
    Z = ad.uns["cnv_linkage"]["linkage"]
    np.testing.assert_array_equal(ad.uns["cnv_linkage"]["leaves"], leaves_list(Z))
    # the cell-level Ward order is its own option; dendrogram=True keeps the reference's meaning (categories)
    axes = cnv.pl.chromosome_heatmap(ad, groupby="g", cell_order="ward", show=False)
    assert "heatmap_ax" in axes and "dendrogram_ax" not in axes
    img = np.asarray(axes["heatmap_ax"].images[0].get_array())
    rank = np.empty(300, dtype=np.int64)
    rank[leaves_list(Z)] = np.arange(300)
    order = np.concatenate([g[np.argsort(rank[g], kind="stable")] for g in (np.arange(0, 100), np.arange(100, 200),
                                                                          np.arange(200, 300))])
    np.testing.assert_allclose(img, X[order].astype(np.float64))
    axes = cnv.pl.chromosome_heatmap(ad, groupby="g", dendrogram=True, show=False)
    assert "dendrogram_ax" in axes


def test_distance_row_blocks_and_single_process_sharded_driver():
    """Row blocks (what each GPU of a sharded job computes) are bit-identical to the rows of the full matrix."""
    import torch
    from infercnvpy_amd import _engine, dist as icd

    X = _blobs(777, 130, 5, seed=3)
    xd = torch.from_numpy(X).cuda()
    full = _engine.pairwise_sqeuclidean(xd)
    for r0, r1 in ((0, 1), (100, 389), (389, 777), (640, 777)):
        blk = _engine.pairwise_sqeuclidean(xd, rows=(r0, r1))
        assert torch.equal(blk, full[r0:r1])
    Z = icd.ward_linkage_sharded(xd)  # world size 1: same code path as tl.ward_linkage
    import infercnvpy_amd as cnv

    np.testing.assert_array_equal(Z, cnv.tl.ward_linkage(X))


@pytest.mark.parametrize("n,d", [(300, 20), (2500, 64), (6000, 48)])
def test_ward_column_layouts_agree(n, d, monkeypatch):
    """Spare-column ("strip") layout with its in-place compaction against the in-place layout: the same entries
    are computed by the same arithmetic, so on data without tied distances the linkage is identical bit for bit.
    Also the row stride n (no spare columns) through the default entry point."""
    import torch
    from infercnvpy_amd import _engine

    X = _blobs(n, d, 9, seed=n + 1)
    xd = torch.from_numpy(X).cuda()
    Zs, rs = _engine.ward_linkage(_engine.pairwise_sqeuclidean(xd, spare=True), spare=True)  # n / 2 spare columns
    monkeypatch.setenv("ICV_WARD_IN_PLACE", "1")
    _knobs()
    Zi, ri = _engine.ward_linkage(_engine.pairwise_sqeuclidean(xd, spare=True), spare=True)
    monkeypatch.delenv("ICV_WARD_IN_PLACE")
    _knobs()
    d2 = torch.empty((n, (n + 3) // 4 * 4), dtype=torch.float32, device="cuda")[:, :n]
    Zn, rn = _engine.ward_linkage(_engine.pairwise_sqeuclidean(xd, out=d2))
    # a column slice of a wider buffer: without the spare flag nothing outside the n x n block is touched
    wide = torch.full((n, 2 * n + 8), 7.0, dtype=torch.float32, device="cuda")
    Zw, rw = _engine.ward_linkage(_engine.pairwise_sqeuclidean(xd, out=wide[:, :n]))
    assert bool((wide[:, n:] == 7.0).all()) and rw == rn
    np.testing.assert_array_equal(Zw, Zn)
    with pytest.raises(ValueError):  # the spare layout needs its stride
        _engine.ward_linkage(_engine.pairwise_sqeuclidean(xd, out=d2), spare=True)
    assert rs == ri == rn
    np.testing.assert_array_equal(Zs, Zi)
    np.testing.assert_array_equal(Zn, Zi)


def test_ward_linkage_properties_at_20000_cells():
    """Size-independent properties where scipy would take minutes: a valid scipy linkage (monotone heights, sizes,
    every cluster id used once), the root holds all cells, and two runs agree bit for bit (deterministic rounds,
    compactions included)."""
    import infercnvpy_amd as cnv
    from scipy.cluster.hierarchy import is_valid_linkage

    n = 20000
    X = _blobs(n, 48, 25, seed=77)
    Z, rounds = cnv.tl.ward_linkage(X, return_rounds=True)
    assert Z.shape == (n - 1, 4) and is_valid_linkage(Z)
    assert np.all(np.diff(Z[:, 2]) >= 0) and Z[-1, 3] == n
    ids = np.concatenate([Z[:, 0], Z[:, 1]]).astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(2 * n - 2))
    assert 10 < rounds < 200
    Z2 = cnv.tl.ward_linkage(X)
    np.testing.assert_array_equal(Z, Z2)


# --------------------------------------------------------------------------- #
# randomized sweep over geometries / dtypes / formats / reference kinds, and the multi-slab driver
# --------------------------------------------------------------------------- #
N_SWEEP = 18
SWEEP_SD_FROM = 12  # seeds from here on: float32 CSR / CSC input in block form = the stored-entries kernel k_smooth_se


def _sweep_case(seed, sd=None):
    rng = np.random.RandomState(1000 + seed)
    n_chr = rng.randint(2, 8)
    if sd is None:
        sd = seed >= SWEEP_SD_FROM
    sizes = [int(rng.choice([17, 99, 100, 101, 180, 333, 400, 700] if sd else [5, 17, 40, 99, 100, 101, 180, 333]))
             for _ in range(n_chr)]
    names = [f"chr{i}" for i in rng.choice(np.arange(1, 23), size=n_chr, replace=False)]
    extra = [("chrX", int(rng.randint(3, 40))), ("chrM", 4), (None, int(rng.randint(1, 6)))][: rng.randint(0, 4)]
    v = cases.synthetic_var(sizes, names=names, extra=tuple(extra), seed_start=seed, seed_perm=seed + 1)
    G = len(v["names"])
    n_obs = int(rng.randint(25, 140))
    kind = rng.choice(["f32", "f64", "int"])
    if sd:
        kind = "f32"
    if kind == "int":
        X = rng.poisson(1.3, size=(n_obs, G)).astype(np.int64)
    else:
        X = cases.synthetic_expr(n_obs, G, seed=seed + 7, dtype=np.float32 if kind == "f32" else np.float64)
    fmt = rng.choice(["dense", "csr", "csc"])
    if sd:
        fmt = ["csr", "csc"][seed % 2]
    labels = rng.choice(["a", "b", "c"], size=n_obs)
    labels[:3] = ["a", "b", "c"]
    kw = dict(window_size=int(rng.choice([4, 6, 10, 20, 50, 100, 101])), step=int(rng.choice([1, 2, 5, 10])),
              lfc_clip=float(rng.choice([0.5, 1.0, 3.0])), chunksize=int(rng.choice([7, 32, 5000])),
              dynamic_threshold=[None, 0.5, 1.5][rng.randint(0, 3)],
              exclude_chromosomes=[("chrX", "chrY"), None, (names[0],)][rng.randint(0, 3)])
    ref_kind = rng.choice(["none", "array", "cat1", "cat2"])
    if sd:  # even windows with a block size gcd(step, window / 2) > 1, clips up to 10, every reference kind
        pairs = [(20, 2), (20, 4), (100, 2), (100, 10), (120, 4), (120, 5), (250, 10), (250, 5), (50, 10), (100, 4)]
        win, stp = pairs[rng.randint(0, len(pairs))]
        kw.update(window_size=win, step=stp, lfc_clip=float(rng.choice([0.5, 3.0, 10.0])))
        ref_kind = ["none", "array", "cat1", "cat2"][seed % 4]
    return v, X, fmt, labels, kw, ref_kind, rng


@pytest.mark.parametrize("seed", range(N_SWEEP))
def test_random_sweep_public_api_against_oracle(seed):
    _check_sweep_case(seed, seed >= SWEEP_SD_FROM)


def _check_sweep_case(seed, sd, extras=False):
    """One random case through the public API against the oracle (also driven by tests/fuzz_gpu.py over more seeds;
    `extras`: a third of the cases each also with calculate_gene_values / as two row shards on the one GPU)."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v, X, fmt, labels, kw, ref_kind, rng = _sweep_case(seed, sd)
    Xin = {"dense": X, "csr": sp.csr_matrix(X), "csc": sp.csc_matrix(X)}[fmt]
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    ad = SimpleAnnData(Xin, obs=pd.DataFrame({"group": labels}), var=var)
    mean_dtype = np.float32 if X.dtype == np.float32 else np.float64
    api = dict(kw)
    if ref_kind == "array":
        ref = (X.mean(axis=0) + rng.normal(0, 0.05, X.shape[1])).astype(mean_dtype)
        api["reference"] = ref
    elif ref_kind == "none":
        ref = _oracle_means(Xin)  # the reference's own mean of the matrix as stored (dense / CSR / CSC orders)
    else:
        cats = ["a"] if ref_kind == "cat1" else ["b", "c"]
        api.update(reference_key="group", reference_cat=cats if len(cats) > 1 else cats[0])
        ref = _oracle_means(Xin, labels, cats)
    tm = {}
    gene_values = bool(extras and rng.rand() < 0.33)
    if extras and rng.rand() < 0.33:
        api["devices"] = [0, 0]
    chr_pos, res, per_gene = cnv.tl.infercnv(ad, inplace=False, _timings=tm, calculate_gene_values=gene_values, **api)
    if sd and not gene_values:  # (per-gene values come from the window array of the generic kernel)
        from infercnvpy_amd import _lib

        assert tm["kernel"] == _lib.ICV_KERNEL_SD, (tm["kernel"], kw)
    e_pos, e_res, e_gene, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref,
                                         calculate_gene_values=gene_values, **kw)
    if gene_values:
        assert per_gene.shape == e_gene.shape
        np.testing.assert_array_equal(np.isnan(per_gene), np.isnan(e_gene))
        gg, ge = np.nan_to_num(per_gene), np.nan_to_num(e_gene)
        gdiffer = (gg == 0) != (ge == 0)
        assert np.all(np.maximum(np.abs(gg), np.abs(ge))[gdiffer] < 1e-13), int(gdiffer.sum())
        np.testing.assert_allclose(gg, ge, rtol=0, atol=ATOL_TIGHT)
    assert {k: int(p) for k, p in chr_pos.items()} == {k: int(p) for k, p in e_pos.items()}
    got, exp = res.toarray(), e_res.toarray()
    assert got.shape == exp.shape
    # The zero pattern is exact, except for entries at the rounding noise of the window sum itself: without a noise
    # threshold a window that TIES with the row's median in exact arithmetic is 0 or ~1e-17 depending on the order of
    # the float64 additions -- and np.convolve's own order depends on the BLAS kernel numpy picks for the host CPU
    # (n <= ~16: sequential; longer: vectorised partial sums).  tests/fuzz_gpu.py finds such ties in ~1 % of the
    # integer-count cases with dynamic_threshold=None; with a threshold those entries are zero on both sides.
    differ = (got == 0) != (exp == 0)
    assert np.all(np.maximum(np.abs(got), np.abs(exp))[differ] < 1e-13), int(differ.sum())
    np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_TIGHT)


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_public_api_multi_slab_equals_single_slab(fmt, monkeypatch):
    """Matrices larger than the HBM budget are processed in row slabs (multiples of chunksize): the
    reference means are accumulated over the slabs and the result must equal the single-slab run."""
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    v = cases.synthetic_var([300, 120, 101, 60])
    X = cases.synthetic_expr(230, len(v["names"]), seed=5)
    labels = np.array(["n1"] * 40 + ["n2"] * 30 + ["t"] * 160)[np.random.RandomState(2).permutation(230)]
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])

    def run(**kw):
        ad = SimpleAnnData(sp.csr_matrix(X) if fmt == "csr" else X, obs=pd.DataFrame({"group": labels}), var=var)
        return cnv.tl.infercnv(ad, inplace=False, chunksize=50, **kw)

    variants = (dict(), dict(reference_key="group", reference_cat=["n1", "n2"]), dict(calculate_gene_values=True))
    single = [run(**kw) for kw in variants]
    from infercnvpy_amd import _engine

    monkeypatch.setattr(_engine, "free_hbm_bytes", lambda: 200_000)  # ~1 chunk per slab
    for kw, (pos1, res1, gv1) in zip(variants, single):
        pos, res, gv = run(**kw)
        assert pos == pos1
        np.testing.assert_array_equal(res.toarray(), res1.toarray())
        if gv1 is not None:
            np.testing.assert_array_equal(gv, gv1)


def test_np_matrix_input():
    """`np.matrix` input (what sparse - dense arithmetic yields; reference `_ensure_array`, _util.py:4-9)."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    v = cases.synthetic_var([120, 60])
    X = cases.synthetic_expr(40, len(v["names"]), seed=3)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    ref = X.mean(axis=0)
    a = cnv.tl.infercnv(SimpleAnnData(X, var=var), reference=ref, window_size=20, step=5, inplace=False)
    b = cnv.tl.infercnv(SimpleAnnData(np.matrix(X), var=var), reference=np.matrix(ref), window_size=20, step=5,
                        inplace=False)
    assert a[0] == b[0]
    np.testing.assert_array_equal(a[1].toarray(), b[1].toarray())


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_many_windows_step1_at_20k_genes(fmt):
    """20 000 genes at step 1 = 17 822 windows: the float64 window array (142 KB) does not fit LDS next to the
    row; the generic kernel keeps it in a per-workgroup HBM line instead (config-1 style call on a full gene set)."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    X = cases.synthetic_expr(12, len(v["names"]), seed=77)
    X[3, 5] = np.nan
    ref = X[4:].mean(axis=0)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    ad = SimpleAnnData(sp.csr_matrix(X) if fmt == "csr" else X, var=var)
    chr_pos, res, gv = cnv.tl.infercnv(ad, reference=ref, window_size=100, step=1, chunksize=5, inplace=False,
                                       calculate_gene_values=(fmt == "dense"))
    e_pos, e_res, e_gv, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref, window_size=100, step=1,
                                       chunksize=5, calculate_gene_values=(fmt == "dense"))
    assert res.shape == (12, 17822) and {k: int(p) for k, p in chr_pos.items()} == {k: int(p) for k, p in e_pos.items()}
    got, exp = res.toarray(), e_res.toarray()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    np.testing.assert_array_equal((got == 0)[ok], (exp == 0)[ok])
    np.testing.assert_allclose(got[ok], exp[ok], rtol=0, atol=ATOL_TIGHT)
    if gv is not None:
        np.testing.assert_allclose(gv, e_gv, rtol=0, atol=1e-9, equal_nan=True)


def test_fresh_process_without_importing_torch_first():
    """Regression: the C ABI library links the HIP runtime by SONAME; PyTorch bundles its own copy under the same
    SONAME.  If this package's library were loaded before torch, the process would end up on the system copy
    and device allocations would fail (hipErrorNoDevice).  `_lib.load()` therefore imports torch first; a
    fresh interpreter that calls the public API straight away must work."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, pandas as pd\n"
        "import infercnvpy_amd as cnv\n"
        "from infercnvpy_amd._compat import SimpleAnnData\n"
        "import cases\n"
        "v = cases.synthetic_var([120, 60])\n"
        "X = cases.synthetic_expr(30, len(v['names']), seed=3)\n"
        "var = pd.DataFrame({'chromosome': v['chromosome'], 'start': v['start'], 'end': v['end']}, index=v['names'])\n"
        "ad = SimpleAnnData(X, var=var)\n"
        "cnv.tl.infercnv(ad, window_size=20, step=5)\n"
        "print('OK', ad.obsm['X_cnv'].shape)\n"
    ) % (os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "golden"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK (30," in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("dtype,genes_per_chrom,fmt", [
    (np.float64, [2600, 2400, 2300, 2200, 2100, 2000, 1900, 1800, 1700, 1600, 1500, 1400, 900, 800], "dense"),  # 25 200
    (np.float64, [2600, 2400, 2300, 2200, 2100, 2000, 1900, 1800, 1700, 1600, 1500, 1400, 900, 800], "csr"),
    (np.float32, [4100, 3900, 3700, 3500, 3300, 3100, 2900, 2700, 2500, 2300, 2100, 1900, 1700, 1500, 1300, 1100,
                  900, 700, 500, 300, 100], "dense"),                                                           # 44 100
    (np.int64, [2600, 2400, 2300, 2200, 2100, 2000, 1900, 1800, 1700, 1600, 1500, 1400, 900, 800], "csr"),
])
def test_gene_sets_larger_than_lds_split_by_chromosome_group(dtype, genes_per_chrom, fmt):
    """More genes than one LDS-resident row holds (float64: ~20 000, float32: ~40 000): the chromosomes are
    processed in groups, median and centring run on the collected float64 windows."""
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var(genes_per_chrom, extra=(("chrX", 50), (None, 7)))
    G = len(v["names"])
    n = 23
    if dtype == np.int64:
        X = np.random.RandomState(4).poisson(0.8, size=(n, G)).astype(np.int64)
    else:
        X = cases.synthetic_expr(n, G, seed=11, dtype=dtype)
    labels = np.array(["a"] * 8 + ["b"] * 7 + ["c"] * 8)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    ad = SimpleAnnData(sp.csr_matrix(X) if fmt == "csr" else X, obs=pd.DataFrame({"group": labels}), var=var)
    gvals = dtype == np.float64 and fmt == "dense"
    mean_dtype = np.float32 if dtype == np.float32 else np.float64
    ref = _oracle_means(ad.X, labels, ["a", "b"])
    assert ref.dtype == mean_dtype
    chr_pos, res, gv = cnv.tl.infercnv(ad, reference_key="group", reference_cat=["a", "b"], chunksize=10,
                                       inplace=False, calculate_gene_values=gvals)
    e_pos, e_res, e_gv, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref, chunksize=10,
                                       calculate_gene_values=gvals)
    assert {k: int(p) for k, p in chr_pos.items()} == {k: int(p) for k, p in e_pos.items()}
    got, exp = res.toarray(), e_res.toarray()
    np.testing.assert_array_equal(got == 0, exp == 0)
    np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_TIGHT)
    if gvals:
        np.testing.assert_allclose(gv, e_gv, rtol=0, atol=1e-9, equal_nan=True)


def test_stored_entries_kernel_gives_the_same_bits_for_any_slot_count():
    """k_smooth_se prefetches PF x 512 stored entries per cell (template parameter, picked from the row-length hint of
    icv_matrix._pad); rows with more take the in-phase loop.  Every choice -- also a hint far too small -- gives the same
    x_res, medians and thresholds bit for bit (and W <= 1536 runs the three-window-register instantiation, > 1536 the
    four-register one)."""
    import torch

    from infercnvpy_amd import _engine, _lib
    from infercnvpy_amd._plan import GenePlan

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    rs = np.random.RandomState(4)
    n = 700
    X = rs.gamma(0.3, 1.0, (n, 20000)).astype(np.float32)
    X[rs.rand(n, 20000) >= 0.07] = 0
    X[3, :] = rs.gamma(0.3, 1.0, 20000)       # a row with 20 000 stored entries
    X[5, 9000:] = 0                             # a short one
    ref = torch.from_numpy(X.mean(axis=0)).cuda()
    for window in (250, 100):                   # 1 472 windows (MAXW = 3) / 1 802 (MAXW = 4)
        plan = GenePlan(v["chromosome"], v["start"], window_size=window, step=10)
        outs = []
        for hint in (None, 1, 600, 1100, 1600, 30000):
            dm = _engine.to_device_matrix(sp.csr_matrix(X))
            if hint is not None:
                dm._row_len_hint = hint
            res = _engine.run_hot_path(plan, dm, ref, chunksize=200, cell_stats=True)
            torch.cuda.synchronize()
            assert plan.last_kernel() == _lib.ICV_KERNEL_SD
            outs.append((res.out.cpu().numpy(), res.cell_median.cpu().numpy(), res.thr.cpu().numpy()))
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                np.testing.assert_array_equal(a, b)
        plan.close()

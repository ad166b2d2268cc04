"""GPU tests (``-m gpu``): the rest of the kept API on the DEVICE-RESIDENT result.

``tl.infercnv`` on a matrix in HBM leaves ``obsm["X_cnv"]`` as a device CSR (``PackedCsr``); the reference's
``cnv_score`` / ``ithcna`` / ``chromosome_heatmap`` read ``obsm["X_cnv"]`` exactly as ``infercnv`` wrote it
(reference tl/_scores.py:65-68, :197-213, pl/_chromosome_heatmap.py:55), so the same chain has to run on that object
and give the host-input chain's answers bit for bit."""
import threading

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import cases

pytestmark = pytest.mark.gpu


def _inputs(n=2300, dtype=np.float32, seed=7):
    v = cases.synthetic_var([700, 320, 260, 150, 100, 60], extra=(("chrX", 40), ("chrM", 5), (None, 3)))
    X = cases.synthetic_expr(n, len(v["names"]), seed=seed, dtype=dtype)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    rs = np.random.RandomState(seed)
    obs = pd.DataFrame({"group": np.array(["n1", "n2", "t"])[rs.randint(0, 3, n)],
                        "clone": np.array(["a", "b", "c", "d", "solo"])[np.minimum(rs.randint(0, 4, n), 3)]},
                       index=[f"c{i}" for i in range(n)])
    obs.loc[obs.index[5], "clone"] = "solo"  # a single-cell group: skipped by ithcna as in the reference
    return X, obs, var


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_chain_on_resident_adata_equals_host_chain_bit_for_bit(fmt):
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd import _engine

    X, obs, var = _inputs()
    Xh = sp.csr_matrix(X) if fmt == "csr" else X
    Xd = _engine.to_device_matrix(Xh) if fmt == "csr" else torch.from_numpy(X).cuda()
    kw = dict(reference_key="group", reference_cat=["n1", "n2"])
    ad_h = cnv.SimpleAnnData(Xh, obs=obs.copy(), var=var)
    ad_d = cnv.SimpleAnnData(Xd, obs=obs.copy(), var=var)
    cnv.tl.infercnv(ad_h, **kw)
    cnv.tl.infercnv(ad_d, **kw)
    assert isinstance(ad_d.obsm["X_cnv"], cnv.PackedCsr) and sp.issparse(ad_h.obsm["X_cnv"])

    # cnv_score: the device arrays are read where they lie, group sums on the device in a fixed order
    s_h = cnv.tl.cnv_score(ad_h, "clone", inplace=False)
    s_d = cnv.tl.cnv_score(ad_d, "clone", inplace=False)
    assert s_h.keys() == s_d.keys()
    for k in s_h:
        assert s_h[k] == s_d[k], (k, s_h[k], s_d[k])
    cnv.tl.cnv_score(ad_h, "clone")
    cnv.tl.cnv_score(ad_d, "clone")
    np.testing.assert_array_equal(ad_h.obs["cnv_score"].values, ad_d.obs["cnv_score"].values)
    # against the definition (reference tl/_scores.py:65-68) on the host matrix
    dense = ad_h.obsm["X_cnv"].toarray()
    for k in s_h:
        assert s_d[k] == pytest.approx(np.mean(np.abs(dense[(obs["clone"] == k).values])), rel=1e-12)

    # ithcna: the group's rows become a float32 tile in HBM (icv_csr_densify) -> the same MFMA contraction.  On the
    # un-thresholded profiles (a cell whose X_cnv row is all zeros has no correlation: NaN, in the reference as well)
    cnv.tl.infercnv(ad_h, key_added="raw", dynamic_threshold=None, **kw)
    cnv.tl.infercnv(ad_d, key_added="raw", dynamic_threshold=None, **kw)
    i_h = cnv.tl.ithcna(ad_h, "clone", use_rep="X_raw", inplace=False)
    i_d = cnv.tl.ithcna(ad_d, "clone", use_rep="X_raw", inplace=False)
    assert "solo" not in i_h and i_h.keys() == i_d.keys() and len(i_h) == 4
    for k in i_h:
        assert np.isfinite(i_h[k]) and i_h[k] == i_d[k], (k, i_h[k], i_d[k])
    n_h = cnv.tl.ithcna(ad_h, "clone", inplace=False)  # thresholded X_cnv: NaN where a group has an all-zero cell
    n_d = cnv.tl.ithcna(ad_d, "clone", inplace=False)
    for k in n_h:
        np.testing.assert_array_equal(n_h[k], n_d[k], err_msg=str(k))

    # cell_linkage (config 5's input never leaves the GPU)
    z_h = cnv.tl.cell_linkage(ad_h, inplace=False)
    z_d = cnv.tl.cell_linkage(ad_d, inplace=False)
    np.testing.assert_array_equal(z_h, z_d)
    cnv.tl.cell_linkage(ad_d)
    np.testing.assert_array_equal(ad_d.uns["cnv_linkage"]["linkage"], z_h)

    # the dense tile is the host matrix's float32 image
    rows = np.flatnonzero((obs["clone"] == "b").values)
    tile = ad_d.obsm["X_cnv"].dense_rows(rows).cpu().numpy()
    np.testing.assert_array_equal(tile, dense[rows].astype(np.float32))
    np.testing.assert_array_equal(ad_d.obsm["X_cnv"].dense_rows((obs["clone"] == "b").values).cpu().numpy(), tile)
    np.testing.assert_array_equal(ad_d.obsm["X_cnv"].toarray(), dense)


def test_reference_answers_pass_through_the_device_csr():
    """Reference tests/test_scores.py:6-21 (cnv_score 2.25 / 2.5, ithcna 1.053 / 0) with X_cnv as a PackedCsr."""
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd import _engine

    x_cnv = np.array([[1, 1, 1, 2, 2, 1, 1, 1], [2, 2, 2, 1, 1, 2, 2, 2], [4, 4, 4, 2, 2, 3, 3, 3],
                      [2, 2, 2, 4, 4, 4, 4, 4]]).T.astype(np.float64)
    m = sp.csr_matrix(x_cnv)
    pk = _engine.PackedCsr(torch.from_numpy(m.indptr.astype(np.int64)).cuda(), torch.from_numpy(m.indices.astype(np.int32)).cuda(),
                           torch.from_numpy(m.data).cuda(), m.shape[1])
    obs = pd.DataFrame({"group": list("AAAAABBB")}, index=[f"c{i}" for i in range(8)])
    ad = cnv.SimpleAnnData(np.zeros((8, 3)), obs=obs, obsm={"X_cnv": pk})
    res = cnv.tl.cnv_score(ad, "group", inplace=False)
    assert res["A"] == 2.25 and res["B"] == 2.5
    cnv.tl.cnv_score(ad, "group")
    np.testing.assert_array_equal(ad.obs["cnv_score"].values, [2.25] * 5 + [2.5] * 3)
    res = cnv.tl.ithcna(ad, "group", inplace=False)
    assert res["A"] == pytest.approx(1.053, abs=1e-3) and res["B"] == 0


def test_heatmap_takes_the_device_csr():
    import matplotlib

    matplotlib.use("Agg")
    import torch

    import infercnvpy_amd as cnv

    X, obs, var = _inputs(n=400)
    ad = cnv.SimpleAnnData(torch.from_numpy(X).cuda(), obs=obs, var=var)
    cnv.tl.infercnv(ad)
    axes = cnv.pl.chromosome_heatmap(ad, groupby="group", show=False)
    assert "heatmap_ax" in axes
    axes = cnv.pl.chromosome_heatmap_summary(ad, groupby="group", show=False)
    assert "heatmap_ax" in axes
    axes = cnv.pl.chromosome_heatmap(ad, groupby="group", show=False, cell_order="ward")
    assert len(ad.uns["cnv_linkage"]["leaves"]) == 400


def test_infercnv_device_is_the_resident_call_without_a_container():
    import torch

    import infercnvpy_amd as cnv

    X, obs, var = _inputs(n=1200)
    Xd = torch.from_numpy(X).cuda()
    pos_h, res_h, gv_h = cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), inplace=False, chunksize=500,
                                         reference_key="group", reference_cat="n1", calculate_gene_values=True)
    pos_d, res_d, gv_d = cnv.tl.infercnv_device(Xd, var, obs, chunksize=500, reference_key="group", reference_cat="n1",
                                                calculate_gene_values=True)
    assert list(pos_d.items()) == list(pos_h.items())
    got = res_d.to_scipy()
    np.testing.assert_array_equal(got.indptr, res_h.indptr)
    np.testing.assert_array_equal(got.indices, res_h.indices)
    np.testing.assert_array_equal(got.data, res_h.data)
    np.testing.assert_array_equal(gv_d.cpu().numpy(), gv_h)
    with pytest.raises(ValueError):
        cnv.tl.infercnv_device(X, var, obs)  # a host matrix
    with pytest.raises(ValueError):
        cnv.tl.infercnv_device(Xd, var.iloc[:-1], obs)
    with pytest.raises(ValueError):
        cnv.tl.infercnv_device(Xd, var, None, reference_key="group", reference_cat="n1")
    with pytest.raises(ValueError):
        cnv.tl.infercnv_device(Xd, var.drop(columns=["start"]), obs)


def test_gene_values_on_a_resident_matrix_that_needs_several_pieces(monkeypatch):
    """calculate_gene_values=True has no size limit in the reference (tl/_infercnv.py:141-151)."""
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd.tl import _infercnv as T

    X, obs, var = _inputs(n=2300)
    ref = X[:300].mean(axis=0)
    Xd = torch.from_numpy(X).cuda()
    pos_1, res_1, gv_1 = cnv.tl.infercnv_device(Xd, var, chunksize=300, reference=ref, calculate_gene_values=True)
    from infercnvpy_amd import _engine

    per_row = 16 * 1400 + 64 + 8 * (X.shape[1] + 1400)
    monkeypatch.setattr(_engine, "free_hbm_bytes", lambda: int(8 * X.shape[1] * 2300 + 700 * per_row / 0.4))
    tm = {}
    pos_k, res_k, gv_k = cnv.tl.infercnv_device(Xd, var, chunksize=300, reference=ref, calculate_gene_values=True,
                                                _timings=tm)
    monkeypatch.undo()
    assert tm["pieces"] > 1
    a, b = res_1.to_scipy(), res_k.to_scipy()
    np.testing.assert_array_equal(a.indptr, b.indptr)
    np.testing.assert_array_equal(a.indices, b.indices)
    np.testing.assert_array_equal(a.data, b.data)
    np.testing.assert_array_equal(gv_1.cpu().numpy(), gv_k.cpu().numpy())
    _, _, gv_h = cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), inplace=False, chunksize=300, reference=ref,
                                 calculate_gene_values=True)
    np.testing.assert_array_equal(gv_k.cpu().numpy(), gv_h)


def test_user_built_device_csr_is_validated():
    import torch

    import infercnvpy_amd as cnv

    m = sp.random(50, 40, density=0.2, format="csr", random_state=0, dtype=np.float32)
    m.sort_indices()
    ip = torch.from_numpy(m.indptr.astype(np.int64)).cuda()
    ix = torch.from_numpy(m.indices.astype(np.int32)).cuda()
    dv = torch.from_numpy(m.data).cuda()
    cnv.DeviceMatrix(indptr=ip, indices=ix, data=dv, shape=m.shape)  # fine
    row = int(np.flatnonzero(np.diff(m.indptr) >= 2)[0])
    e = int(m.indptr[row])
    swapped = m.indices.astype(np.int32).copy()
    swapped[e], swapped[e + 1] = swapped[e + 1], swapped[e]
    with pytest.raises(ValueError, match="ascending"):
        cnv.DeviceMatrix(indptr=ip, indices=torch.from_numpy(swapped).cuda(), data=dv, shape=m.shape)
    dup = m.indices.astype(np.int32).copy()
    dup[e + 1] = dup[e]
    with pytest.raises(ValueError, match="ascending"):
        cnv.DeviceMatrix(indptr=ip, indices=torch.from_numpy(dup).cuda(), data=dv, shape=m.shape)
    big = m.indices.astype(np.int32).copy()
    big[e] = 40
    with pytest.raises(ValueError, match="column index"):
        cnv.DeviceMatrix(indptr=ip, indices=torch.from_numpy(big).cuda(), data=dv, shape=m.shape)
    bad_ip = m.indptr.astype(np.int64).copy()
    bad_ip[-1] += 5
    with pytest.raises(ValueError, match="offsets"):
        cnv.DeviceMatrix(indptr=torch.from_numpy(bad_ip).cuda(), indices=ix, data=dv, shape=m.shape)
    with pytest.raises(ValueError):
        cnv.DeviceMatrix(indptr=ip.to(torch.int32), indices=ix, data=dv, shape=m.shape)
    with pytest.raises(ValueError):
        cnv.DeviceMatrix(indptr=ip, indices=ix, data=dv.to(torch.float16), shape=m.shape)
    with pytest.raises(ValueError):
        cnv.DeviceMatrix(indptr=ip[:-1], indices=ix, data=dv, shape=m.shape)
    with pytest.raises(ValueError):
        cnv.DeviceMatrix(dense=torch.zeros((4, 4), device="cuda", dtype=torch.float16))


def test_mean_order_float64_is_the_correctly_rounded_mean():
    import torch

    import infercnvpy_amd as cnv

    X, obs, var = _inputs(n=3000)
    ref64 = X.astype(np.float64).mean(axis=0).astype(np.float32)
    _, exp, _ = cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), inplace=False, reference=ref64, chunksize=500)
    for Xin, kw in ((X, dict()), (X, dict(devices=[0, 0, 0])), (torch.from_numpy(X).cuda(), dict())):
        _, got, _ = cnv.tl.infercnv(cnv.SimpleAnnData(Xin, obs=obs, var=var), inplace=False, mean_order="float64",
                                    chunksize=500, **kw)
        got = got.to_scipy() if isinstance(got, cnv.PackedCsr) else got
        np.testing.assert_array_equal(got.indptr, exp.indptr)
        np.testing.assert_array_equal(got.indices, exp.indices)
        np.testing.assert_array_equal(got.data, exp.data)
    # per-category means the same way
    sel = (obs["group"] == "n2").values
    refc = X[sel].astype(np.float64).mean(axis=0).astype(np.float32)
    _, exp, _ = cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), inplace=False, reference=refc, chunksize=500)
    for Xin in (sp.csr_matrix(X), torch.from_numpy(X).cuda()):
        _, got, _ = cnv.tl.infercnv(cnv.SimpleAnnData(Xin, obs=obs, var=var), inplace=False, mean_order="float64",
                                    chunksize=500, reference_key="group", reference_cat="n2")
        got = got.to_scipy() if isinstance(got, cnv.PackedCsr) else got
        np.testing.assert_array_equal(got.indices, exp.indices)
        np.testing.assert_array_equal(got.data, exp.data)
    with pytest.raises(ValueError):
        cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), mean_order="fast")


def test_concurrent_resident_calls_each_get_a_plan():
    """a shared cached plan answered the second thread with 'plan busy'."""
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd.tl import _infercnv as T

    T._clear_plan_cache()
    X, obs, var = _inputs(n=6000)
    Xd = torch.from_numpy(X).cuda()
    ref = X[:300].mean(axis=0)
    _, first, _ = cnv.tl.infercnv_device(Xd, var, reference=ref)
    first = first.to_scipy()
    errs, outs = [], []

    def work():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(6):
                    outs.append(cnv.tl.infercnv_device(Xd, var, reference=ref)[1])
                torch.cuda.current_stream().synchronize()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work) for _ in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for o in outs:
        o = o.to_scipy()
        np.testing.assert_array_equal(o.indptr, first.indptr)
        np.testing.assert_array_equal(o.data, first.data)
    ent = next(iter(T._PLAN_CACHE.values()))
    assert ent.busy == 0 and 1 <= len(ent.idle) <= 4
    # eviction closes idle plans only; an annotation edited in place is a new key (content, not object identity)
    var2 = var.copy()
    var2.loc[var2.index[0], "chromosome"] = "chr2"
    cnv.tl.infercnv_device(Xd, var2, reference=ref)
    assert len(T._PLAN_CACHE) == 2
    T._clear_plan_cache()
    assert not T._PLAN_CACHE


def test_a_real_anndata_gets_host_objects():
    """anndata validates what goes into obsm / layers and takes neither a PackedCsr nor a device tensor:
    handed a container of the anndata package, tl.infercnv writes the reference's host objects."""
    import torch

    import infercnvpy_amd as cnv

    class AnnData(cnv.SimpleAnnData):  # stands in for anndata.AnnData (the package is not installed here)
        pass

    AnnData.__module__ = "anndata._core.anndata"
    X, obs, var = _inputs(n=900)
    ad = AnnData(torch.from_numpy(X).cuda(), obs=obs, var=var)
    cnv.tl.infercnv(ad, chunksize=300, calculate_gene_values=True)
    assert sp.issparse(ad.obsm["X_cnv"]) and ad.obsm["X_cnv"].dtype == np.float64
    assert isinstance(ad.layers["gene_values_cnv"], np.ndarray)
    ref = cnv.SimpleAnnData(X, obs=obs, var=var)
    cnv.tl.infercnv(ref, chunksize=300, calculate_gene_values=True)
    np.testing.assert_array_equal(ad.obsm["X_cnv"].toarray(), ref.obsm["X_cnv"].toarray())
    np.testing.assert_array_equal(ad.layers["gene_values_cnv"], ref.layers["gene_values_cnv"])

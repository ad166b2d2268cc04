"""GPU tests (``-m gpu``): ``tl.infercnv`` on a matrix that already lives in HBM (a CUDA tensor / DeviceMatrix as
``adata.X``) gives the host-input call's X_cnv bit for bit, without copies or synchronisation, and keeps its plan."""
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
    labels = np.array(["n1", "n2", "t"])[np.random.RandomState(seed).randint(0, 3, n)]
    obs = pd.DataFrame({"group": labels}, index=[f"c{i}" for i in range(n)])
    return X, obs, var


def _same_csr(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype
    np.testing.assert_array_equal(a.indptr, b.indptr)
    np.testing.assert_array_equal(a.indices, b.indices)
    np.testing.assert_array_equal(a.data, b.data)


@pytest.mark.parametrize("fmt", ["dense", "csr"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_device_input_equals_host_input_bit_for_bit(fmt, dtype):
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd import _engine
    from infercnvpy_amd._compat import SimpleAnnData

    X, obs, var = _inputs(dtype=dtype)
    Xh = sp.csr_matrix(X) if fmt == "csr" else X
    Xd = _engine.to_device_matrix(Xh) if fmt == "csr" else torch.from_numpy(X).cuda()
    ref = X[:300].mean(axis=0).astype(dtype)
    variants = [dict(), dict(reference=ref), dict(reference_key="group", reference_cat=["n2", "n1"]),
                dict(reference_key="group", reference_cat="n1", window_size=250, chunksize=300),
                dict(reference=np.vstack([ref, ref * 0.5]).astype(dtype), dynamic_threshold=None, lfc_clip=1.0),
                dict(reference=ref, calculate_gene_values=True, chunksize=500)]
    for kw in variants:
        pos_h, res_h, gv_h = cnv.tl.infercnv(SimpleAnnData(Xh, obs=obs, var=var), inplace=False, **kw)
        pos_d, res_d, gv_d = cnv.tl.infercnv(SimpleAnnData(Xd, obs=obs, var=var), inplace=False, **kw)
        assert isinstance(res_d, cnv.PackedCsr) and res_d.shape == res_h.shape
        assert list(pos_d.items()) == list(pos_h.items())
        _same_csr(res_d.to_scipy(), res_h)
        if gv_h is not None:
            np.testing.assert_array_equal(gv_d.cpu().numpy(), gv_h)
    # in place: the same fields as the host call
    ad = SimpleAnnData(Xd, obs=obs, var=var)
    assert cnv.tl.infercnv(ad, reference=ref, key_added="k") is None
    assert isinstance(ad.obsm["X_k"], cnv.PackedCsr) and set(ad.uns["k"]) == {"chr_pos"}
    with pytest.raises(ValueError):
        cnv.tl.infercnv(SimpleAnnData(Xd, obs=obs, var=var), reference=ref[:-1])
    with pytest.raises(ValueError):
        cnv.tl.infercnv(SimpleAnnData(Xd, obs=obs, var=var), reference_key="group", reference_cat="nope")


def test_resident_calls_reuse_their_plan_and_do_not_synchronise():
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from infercnvpy_amd.tl import _infercnv as T

    T._clear_plan_cache()
    X, obs, var = _inputs(n=20000)
    Xd = torch.from_numpy(X).cuda()
    ad = SimpleAnnData(Xd, obs=obs, var=var)
    cnv.tl.infercnv(ad)
    n_plans = len(T._PLAN_CACHE)
    first = ad.obsm["X_cnv"].to_scipy()
    plan = T._cached_plan(var["chromosome"].to_numpy(), var["start"].to_numpy(), 100, 10, ("chrX", "chrY"), 0)
    for _ in range(5):
        cnv.tl.infercnv(ad)
    assert len(T._PLAN_CACHE) == n_plans
    assert T._cached_plan(var["chromosome"].to_numpy(), var["start"].to_numpy(), 100, 10, ("chrX", "chrY"), 0) is plan
    _same_csr(ad.obsm["X_cnv"].to_scipy(), first)
    cnv.tl.infercnv(ad, window_size=250)
    assert len(T._PLAN_CACHE) == n_plans + 1
    # another annotation with the same content (new string objects) is a new key at worst, never a wrong plan
    var2 = var.copy()
    var2["chromosome"] = [None if c is None else str(c) + "" for c in var["chromosome"]]
    ad2 = SimpleAnnData(Xd, obs=obs, var=var2)
    cnv.tl.infercnv(ad2)
    _same_csr(ad2.obsm["X_cnv"].to_scipy(), first)


def test_host_call_leaves_nothing_for_the_cyclic_collector():
    """A finished call's helpers (CsrDrain, SlabStream, packers) are freed by reference counting: a reference cycle kept
    the drain -- and with it the host arrays of X_cnv, gigabytes at scale -- alive until the cyclic collector ran, which
    then stalled an unrelated later call by 0.15 s (profiles/r06_gc_stall.txt)."""
    import gc

    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    X, obs, var = _inputs(n=3000)
    ref = X[:100].mean(axis=0)
    for mat in (X, sp.csr_matrix(np.where(X > np.quantile(X, 0.8), X, 0)), torch.from_numpy(X).cuda()):
        cnv.tl.infercnv(SimpleAnnData(mat, obs=obs, var=var))  # (first call: one-time objects)
        gc.collect()
        gc.set_debug(gc.DEBUG_SAVEALL)
        try:
            gc.garbage.clear()
            ad = SimpleAnnData(mat, obs=obs, var=var)
            cnv.tl.infercnv(ad, chunksize=500)
            cnv.tl.infercnv(ad, reference=ref, chunksize=500, calculate_gene_values=True)
            del ad
            gc.collect()
            ours = [type(o).__name__ for o in gc.garbage if (type(o).__module__ or "").startswith("infercnvpy_amd")]
        finally:
            gc.set_debug(0)
            gc.garbage.clear()
        assert not ours, ours

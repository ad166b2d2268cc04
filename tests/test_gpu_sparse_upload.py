"""GPU tests (``-m gpu``): a mostly-zero DENSE host matrix crosses PCIe as its stored entries (host threads pack
``indptr / indices / values``: ``icv_host_dense_row_nnz`` + ``icv_host_dense_pack``) and is rebuilt as the same dense
rows in HBM (``icv_csr_scatter_dense``) -- ``X_cnv`` is bit-identical to the plain dense upload (reference
tl/_infercnv.py:115-116, :422-423: the reference densifies every chunk on the host)."""
import numpy as np
import pandas as pd
import pytest

import cases

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert a.shape == b.shape
    np.testing.assert_array_equal(a.indptr, b.indptr)
    np.testing.assert_array_equal(a.indices, b.indices)
    np.testing.assert_array_equal(a.data, b.data)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sparse_upload_equals_dense_upload_bit_for_bit(dtype, monkeypatch):
    import infercnvpy_amd as cnv
    from infercnvpy_amd import _engine

    v = cases.synthetic_var([700, 320, 260, 150, 100, 60], extra=(("chrX", 40), ("chrM", 5), (None, 3)))
    n = 2300
    X = cases.synthetic_expr(n, len(v["names"]), seed=7, dtype=dtype)  # ~19 % non-zeros, 1638 columns
    X[5] = 1.25        # a row without zeros
    X[9] = 0           # an empty row
    X[11, 3] = np.nan  # a stored NaN stays a NaN
    X[12, 4] = -0.0
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    labels = np.array(["n1", "n2", "t"])[np.random.RandomState(7).randint(0, 3, n)]
    obs = pd.DataFrame({"group": labels}, index=[f"c{i}" for i in range(n)])
    assert _engine._wants_sparse_upload(X, dtype, 0, n)
    variants = [dict(), dict(reference_key="group", reference_cat=["n2", "n1"], chunksize=300),
                dict(reference=X[:300].mean(axis=0), window_size=250, calculate_gene_values=True, chunksize=500),
                dict(devices=[0, 0, 0], chunksize=300)]
    for kw in variants:
        tm = {}
        _, sparse_way, gv_s = cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), inplace=False, _timings=tm, **kw)
        shards = tm.get("shards") or [tm]
        assert all(s.get("sparse_upload") for s in shards), tm
        monkeypatch.setenv("ICV_NO_SPARSE_UPLOAD", "1")
        tm = {}
        _, dense_way, gv_d = cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), inplace=False, _timings=tm, **kw)
        monkeypatch.delenv("ICV_NO_SPARSE_UPLOAD")
        assert not any(s.get("sparse_upload") for s in (tm.get("shards") or [tm]))
        _same(sparse_way, dense_way)
        if gv_s is not None:
            np.testing.assert_array_equal(gv_s, gv_d)


def test_sparse_upload_is_chosen_only_where_it_pays():
    from infercnvpy_amd import _engine

    rs = np.random.RandomState(0)
    X = rs.gamma(0.3, 1.0, (400, 2000)).astype(np.float32)
    assert not _engine._wants_sparse_upload(X, np.float32, 0, 400)             # no zeros at all
    X[X < 0.5] = 0
    assert _engine._wants_sparse_upload(X, np.float32, 0, 400)
    assert not _engine._wants_sparse_upload(X, np.float64, 0, 400)             # would need a dtype conversion
    assert not _engine._wants_sparse_upload(X[:, :500], np.float32, 0, 400)    # too narrow to matter
    assert not _engine._wants_sparse_upload(np.asfortranarray(X), np.float32, 0, 400)  # rows are not contiguous
    assert _engine._wants_sparse_upload(X[:, :1500], np.float32, 0, 400)       # a column slice: padded rows are fine
    assert not _engine._wants_sparse_upload(X[:, ::2], np.float32, 0, 400)
    import scipy.sparse as sp

    assert not _engine._wants_sparse_upload(sp.csr_matrix(X), np.float32, 0, 400)


def test_an_error_leaves_no_threads(monkeypatch):
    """The packer / copier pair of a slab is shut down on every exit path."""
    import threading

    import infercnvpy_amd as cnv
    from infercnvpy_amd import _engine, _lib

    v = cases.synthetic_var([700, 320, 260])
    n = 3000
    X = cases.synthetic_expr(n, len(v["names"]), seed=3)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    before = threading.active_count()
    ref = X[:100].mean(axis=0)
    _, a, _ = cnv.tl.infercnv(cnv.SimpleAnnData(X, var=var), inplace=False, reference=ref, chunksize=100)
    calls = []

    def boom(*args, **kw):
        calls.append(1)
        raise RuntimeError("injected")

    monkeypatch.setattr(_engine, "threshold_csr", boom)
    with pytest.raises(RuntimeError, match="injected"):
        cnv.tl.infercnv(cnv.SimpleAnnData(np.vstack([X] * 30), var=var), inplace=False, reference=ref, chunksize=100)
    monkeypatch.undo()
    import time

    for _ in range(50):
        if threading.active_count() <= before + 3:  # (the pinned D2H ring keeps its three helper threads)
            break
        time.sleep(0.1)
    assert threading.active_count() <= before + 3
    _, b, _ = cnv.tl.infercnv(cnv.SimpleAnnData(X, var=var), inplace=False, reference=ref, chunksize=100)
    _same(a, b)

"""The tool functions on a real ``anndata.AnnData`` (skipped where anndata is not installed, as in the build
container): categorical ``var["chromosome"]``, nullable-integer ``start``, a layer, the written fields.
Reference: ``tl/_infercnv.py:110,153-158`` (``adata[:, mask]`` view, ``obsm`` / ``uns`` / ``layers`` writes)."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import cases

anndata = pytest.importorskip("anndata")
pytestmark = pytest.mark.gpu


def test_real_anndata_round_trip(tmp_path):
    import infercnvpy_amd as cnv
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var([300, 150, 101, 60], extra=(("chrX", 20), (None, 3)))
    n = len(v["names"])
    X = cases.synthetic_expr(120, n, seed=9)
    var = pd.DataFrame({
        "chromosome": pd.Categorical(v["chromosome"]),                    # categorical, with missing values
        "start": pd.array(v["start"], dtype="Int64"),                     # nullable integer
        "end": pd.array(v["end"], dtype="Int64"),
    }, index=v["names"])
    obs = pd.DataFrame({"cell_type": pd.Categorical(np.where(np.arange(120) < 50, "normal", "tumor"))},
                       index=[f"c{i}" for i in range(120)])
    ad = anndata.AnnData(X=sp.csr_matrix(X), obs=obs, var=var)
    ad.layers["copy"] = ad.X.copy()
    cnv.tl.infercnv(ad, reference_key="cell_type", reference_cat="normal", chunksize=40)
    pos, exp, _, _ = O.infercnv(X, v["chromosome"], v["start"], obs_col=obs["cell_type"].values,
                                reference_cat=["normal"], chunksize=40)
    got = ad.obsm["X_cnv"]
    assert sp.issparse(got) and got.dtype == np.float64 and got.shape == exp.shape
    assert {k: int(x) for k, x in ad.uns["cnv"]["chr_pos"].items()} == {k: int(x) for k, x in pos.items()}
    assert np.mean((got.toarray() == 0) != (exp.toarray() == 0)) < 2e-3  # means computed on the GPU: ulp ties
    np.testing.assert_allclose(got.toarray()[(got.toarray() != 0) & (exp.toarray() != 0)],
                               exp.toarray()[(got.toarray() != 0) & (exp.toarray() != 0)], atol=1e-5)
    # layer= gives the same result as X (reference tests/test_tools.py:221-239)
    ad2 = ad.copy()
    cnv.tl.infercnv(ad2, reference_key="cell_type", reference_cat="normal", chunksize=40, layer="copy",
                    key_added="cnv2")
    assert (ad2.obsm["X_cnv2"] != ad.obsm["X_cnv"]).nnz == 0
    # h5ad round trip keeps what the plots need
    ad.obs["cnv_leiden"] = pd.Categorical(np.where(np.arange(120) % 2 == 0, "0", "1"))
    cnv.tl.cnv_score(ad)
    path = tmp_path / "a.h5ad"
    try:
        ad.write_h5ad(path)
    except ImportError:
        return
    back = anndata.read_h5ad(path)
    import matplotlib

    matplotlib.use("Agg")
    assert "heatmap_ax" in cnv.pl.chromosome_heatmap(back, show=False)
    assert "heatmap_ax" in cnv.pl.chromosome_heatmap_summary(back, show=False)

"""pl.chromosome_heatmap / pl.chromosome_heatmap_summary (reference tests/test_plotting.py:4-9 are smoke tests on
the real dataset; here the same calls run on a synthetic X_cnv, plus the contract the reference gets from scanpy:
category order, dendrogram over the categories, chromosome boundary lines).  Host-only: no GPU needed."""
import matplotlib

matplotlib.use("Agg")
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import infercnvpy_amd as cnv
from infercnvpy_amd._compat import SimpleAnnData


def _adata(categorical=True, sparse=True, seed=0):
    rng = np.random.RandomState(seed)
    n, w = 60, 300
    groups = np.array(["tumor"] * 25 + ["normal"] * 20 + ["b cell"] * 15)[rng.permutation(n)]
    x = rng.standard_normal((n, w)) * 0.02
    x[groups == "tumor", 40:120] += 0.3
    x[groups == "b cell", 200:260] -= 0.25
    x[np.abs(x) < 0.03] = 0
    obs = pd.DataFrame({"cell_type": groups}, index=[f"c{i}" for i in range(n)])
    if categorical:  # category order differs from the sorted order on purpose
        obs["cell_type"] = pd.Categorical(groups, categories=["tumor", "b cell", "normal"])
    var = pd.DataFrame(index=[f"g{i}" for i in range(5)])
    ad = SimpleAnnData(np.zeros((n, 5), dtype=np.float32), obs=obs, var=var)
    ad.obsm["X_cnv"] = sp.csr_matrix(x) if sparse else x
    ad.uns["cnv"] = {"chr_pos": {"chr2": 90, "chr1": 0, "chr3": 210}}  # order lost on purpose (reference :57-59)
    return ad, x, groups


def _vline_xs(ax):
    xs = []
    for coll in ax.collections:
        for seg in coll.get_segments():
            if len(seg) == 2 and seg[0][0] == seg[1][0]:
                xs.append(float(seg[0][0]))
    return sorted(set(xs))


@pytest.mark.parametrize("sparse", [True, False])
def test_plot_chromosome_heatmap(sparse):
    ad, x, groups = _adata(sparse=sparse)
    axes = cnv.pl.chromosome_heatmap(ad, groupby="cell_type", show=False)
    assert {"heatmap_ax", "groupby_ax", "gene_groups_ax"} <= set(axes) and "dendrogram_ax" not in axes
    assert _vline_xs(axes["heatmap_ax"]) == [90.0, 210.0]  # chr_pos[1:], re-sorted by value
    img = np.asarray(axes["heatmap_ax"].images[0].get_array())
    assert img.shape == x.shape
    # rows grouped in CATEGORY order (tumor, b cell, normal), original order inside a group
    order = np.concatenate([np.flatnonzero(groups == g) for g in ("tumor", "b cell", "normal")])
    np.testing.assert_allclose(img, x[order])


def test_plot_chromosome_heatmap_sorted_labels_without_categories():
    ad, x, groups = _adata(categorical=False)
    axes = cnv.pl.chromosome_heatmap(ad, groupby="cell_type", show=False)
    order = np.concatenate([np.flatnonzero(groups == g) for g in sorted(set(groups))])
    np.testing.assert_allclose(np.asarray(axes["heatmap_ax"].images[0].get_array()), x[order])


def test_plot_chromosome_heatmap_summary():
    ad, x, groups = _adata()
    axes = cnv.pl.chromosome_heatmap_summary(ad, groupby="cell_type", show=False)
    img = np.asarray(axes["heatmap_ax"].images[0].get_array())
    assert img.shape == (30, x.shape[1])  # 10 rows per group (reference :156-158)
    for k, g in enumerate(("tumor", "b cell", "normal")):
        for r in range(10):
            np.testing.assert_allclose(img[10 * k + r], x[groups == g].mean(axis=0))
    assert _vline_xs(axes["heatmap_ax"]) == [90.0, 210.0]


def test_dendrogram_is_over_categories_like_scanpy():
    from scipy.cluster import hierarchy as sch
    from scipy.spatial import distance

    ad, x, groups = _adata()
    axes = cnv.pl.chromosome_heatmap(ad, groupby="cell_type", dendrogram=True, show=False)
    assert "dendrogram_ax" in axes and len(axes["dendrogram_ax"].lines) == 2  # 3 categories: 2 merges
    cats = ["tumor", "b cell", "normal"]
    means = np.vstack([x[groups == c].mean(axis=0) for c in cats])
    z = sch.linkage(distance.squareform(1 - np.corrcoef(means), checks=False), method="complete")
    leaves = sch.dendrogram(z, no_plot=True)["leaves"]
    order = np.concatenate([np.flatnonzero(groups == cats[i]) for i in leaves])
    np.testing.assert_allclose(np.asarray(axes["heatmap_ax"].images[0].get_array()), x[order])


def test_default_groupby_requires_leiden():
    ad, _, _ = _adata()
    with pytest.raises(ValueError):
        cnv.pl.chromosome_heatmap(ad, show=False)
    with pytest.raises(ValueError):
        cnv.pl.chromosome_heatmap_summary(ad, show=False)
    with pytest.raises(ValueError):
        cnv.pl.chromosome_heatmap(ad, groupby="cell_type", cell_order="nope", show=False)

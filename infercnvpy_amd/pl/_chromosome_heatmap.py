"""``pl.chromosome_heatmap`` / ``pl.chromosome_heatmap_summary``.

Host-side glue with the reference's signatures (icbi-lab/infercnvpy
``src/infercnvpy/pl/_chromosome_heatmap.py:11-193``).  The reference forwards to
``scanpy.pl.heatmap``; scanpy is not a dependency of this package, so an equivalent matplotlib
figure is drawn directly (cells grouped by ``groupby`` in category order, colour map centred at 0 with
``TwoSlopeNorm``, chromosome boundaries as vertical lines, chromosome labels on top).  The returned
dict has scanpy's keys: ``heatmap_ax``, ``groupby_ax``, ``gene_groups_ax`` and, with ``dendrogram=True``,
``dendrogram_ax``.

``dendrogram=True`` means what it means in the reference (kwarg forwarded to ``sc.pl.heatmap``, :74-85, which
runs ``sc.tl.dendrogram``): a dendrogram over the CATEGORIES of ``groupby`` -- per-category means, Pearson
correlation, complete linkage on ``1 - corr`` -- drawn next to the heatmap, categories reordered by its leaves.
(scanpy takes the means over a 50-component PCA of the matrix when it has more than 50 columns; here they are
taken over ``X_cnv`` itself.  The reference pins neither: parity unpinned)
The cell-level Ward ordering of BASELINE config 5 is a separate, explicit option: ``cell_order="ward"`` orders
the cells inside every group by the leaves of ``tl.cell_linkage`` (GPU pairwise distances + Ward).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def _host_matrix(x):
    """``obsm["X_cnv"]`` as the plots read it: a device-resident ``PackedCsr`` (tl.infercnv on a matrix in HBM) is copied
    to a host scipy CSR matrix here, at the plot boundary (reference :55 reads the matrix as infercnv wrote it)."""
    return x.to_scipy() if hasattr(x, "to_scipy") and hasattr(x, "dense_rows") else x


def _sorted_chr_pos(adata, use_rep):
    # re-sort: saving and loading AnnData does not keep dict order (reference :57-59)
    items = sorted(adata.uns[use_rep]["chr_pos"].items(), key=lambda kv: kv[1])
    return [k for k, _ in items], [int(v) for _, v in items]


def _categories(labels):
    """Group order as scanpy plots it: the categories of a categorical column, else the sorted unique labels."""
    cat = getattr(labels, "cat", None)
    if cat is not None:  # pandas Series of categorical dtype
        present = set(np.asarray(labels).tolist())
        return [c for c in cat.categories if c in present]
    if hasattr(labels, "categories"):  # pandas Categorical
        present = set(np.asarray(labels).tolist())
        return [c for c in labels.categories if c in present]
    cats = list(dict.fromkeys(np.asarray(labels).tolist()))
    try:
        return sorted(cats)
    except TypeError:
        return cats


def _category_dendrogram(matrix, labels, cats):
    """scanpy's sc.tl.dendrogram defaults on the category means: Pearson correlation, complete linkage."""
    from scipy.cluster import hierarchy as sch
    from scipy.spatial import distance

    means = np.vstack([np.asarray(matrix[labels == c].mean(axis=0)).ravel() for c in cats])
    with np.errstate(invalid="ignore", divide="ignore"):
        corr = np.corrcoef(means)
    corr = np.nan_to_num(np.atleast_2d(corr), nan=0.0)
    np.fill_diagonal(corr, 1.0)
    cond = distance.squareform(np.clip(1.0 - corr, 0.0, None), checks=False)
    z = sch.linkage(cond, method="complete")
    return z, sch.dendrogram(z, no_plot=True)


def _draw(matrix, labels, chr_names, chr_pos, *, groupby, cmap, figsize, vmin, vmax, show, save, ymin_lines,
          dendrogram=False, cell_rank=None, **kwargs):
    import matplotlib.pyplot as plt
    from matplotlib.colors import TwoSlopeNorm

    cats = _categories(labels)
    labels = np.asarray(labels)
    dendro = None
    if dendrogram and len(cats) > 1:
        _, dendro = _category_dendrogram(matrix, labels, cats)
        cats = [cats[i] for i in dendro["leaves"]]
    order = np.concatenate([np.flatnonzero(labels == c) for c in cats]) if len(cats) else np.arange(0)
    if cell_rank is not None:  # cells of a group in Ward leaf order (cell_order="ward")
        order = np.concatenate([g[np.argsort(cell_rank[g], kind="stable")]
                                for g in (np.flatnonzero(labels == c) for c in cats)]) if len(cats) else order
    mat = matrix[order]
    sizes = [int((labels == c).sum()) for c in cats]

    norm = kwargs.pop("norm", None) or TwoSlopeNorm(0, vmin=vmin, vmax=vmax)
    fig = plt.figure(figsize=figsize)
    ncol = 4 if dendro is not None else 3
    widths = [0.3, 16, 0.8, 0.25] if dendro is not None else [0.3, 16, 0.25]
    gs = fig.add_gridspec(2, ncol, width_ratios=widths, height_ratios=[0.25, 10], wspace=0.02, hspace=0.02)
    ax_groups = fig.add_subplot(gs[1, 0])
    ax_heat = fig.add_subplot(gs[1, 1])
    ax_chr = fig.add_subplot(gs[0, 1], sharex=ax_heat)
    ax_cbar = fig.add_subplot(gs[1, ncol - 1])
    ax_dendro = fig.add_subplot(gs[1, 2]) if dendro is not None else None

    im = ax_heat.imshow(mat, aspect="auto", cmap=cmap, norm=norm, interpolation="nearest",
                        extent=(0, mat.shape[1], mat.shape[0], 0))
    ax_heat.set_xticks([])
    ax_heat.set_yticks([])
    bounds = np.cumsum([0] + sizes)
    for b in bounds[1:-1]:
        ax_heat.axhline(b, color="black", lw=0.5)
    ax_heat.vlines(chr_pos[1:], lw=0.6, ymin=ymin_lines, ymax=mat.shape[0], color="black")

    group_colors = plt.get_cmap("tab20")(np.arange(len(cats)) % 20)
    for i, c in enumerate(cats):
        ax_groups.axhspan(bounds[i], bounds[i + 1], color=group_colors[i])
        ax_groups.text(-0.1, (bounds[i] + bounds[i + 1]) / 2, str(c), ha="right", va="center", fontsize=8,
                       transform=ax_groups.get_yaxis_transform())
    ax_groups.set_ylim(mat.shape[0], 0)
    ax_groups.set_xticks([])
    ax_groups.set_yticks([])
    ax_groups.set_ylabel(groupby)

    ends = chr_pos[1:] + [mat.shape[1]]
    for name, a, b in zip(chr_names, chr_pos, ends):
        ax_chr.plot([a, b], [0, 0], color="black", lw=1)
        ax_chr.text((a + b) / 2, 0.3, name.replace("chr", ""), ha="center", va="bottom", fontsize=7)
    ax_chr.set_ylim(-0.5, 2)
    ax_chr.axis("off")
    fig.colorbar(im, cax=ax_cbar)

    axes = {"heatmap_ax": ax_heat, "groupby_ax": ax_groups, "gene_groups_ax": ax_chr}
    if ax_dendro is not None:
        # scipy puts leaf i at 5 + 10 i: map the leaves onto the centres of the (unequal) category bands
        centres = (bounds[:-1] + bounds[1:]) / 2.0
        leaf_x = 5.0 + 10.0 * np.arange(len(cats))
        for xs, ys in zip(dendro["icoord"], dendro["dcoord"]):
            ax_dendro.plot(ys, np.interp(xs, leaf_x, centres), color="#555555", lw=0.8)
        ax_dendro.set_ylim(mat.shape[0], 0)
        ax_dendro.axis("off")
        axes["dendrogram_ax"] = ax_dendro
    if save:
        fname = save if isinstance(save, str) else "heatmap.png"
        fig.savefig(fname, bbox_inches="tight")
    if show:
        plt.show()
        return None
    return axes


def chromosome_heatmap(adata, *, groupby: str = "cnv_leiden", use_rep: str = "cnv", cmap="bwr",
                       figsize=(16, 10), show=None, save=None, **kwargs):
    """Heatmap of smoothed gene expression by chromosome (reference :11-92)."""
    if groupby == "cnv_leiden" and "cnv_leiden" not in adata.obs.columns:
        raise ValueError("'cnv_leiden' is not in `adata.obs`. Did you run `tl.leiden()`?")
    x_dev = adata.obsm[f"X_{use_rep}"]
    x = _host_matrix(x_dev)
    chr_names, chr_pos = _sorted_chr_pos(adata, use_rep)

    data = x.data if sp.issparse(x) else np.asarray(x)
    vmin = kwargs.pop("vmin", None)
    vmax = kwargs.pop("vmax", None)
    if vmin is None:
        vmin = np.nanmin(data)
    if vmax is None:
        vmax = np.nanmax(data)

    dense = x.toarray() if sp.issparse(x) else np.asarray(x)
    cell_rank = None
    cell_order = kwargs.pop("cell_order", None)
    if cell_order is not None:
        if cell_order != "ward":
            raise ValueError("cell_order must be None or 'ward'")
        from ..tl._linkage import cell_linkage

        key = f"{use_rep}_linkage"
        if key not in adata.uns or len(adata.uns[key]["leaves"]) != dense.shape[0]:
            cell_linkage(adata, use_rep=use_rep)
        cell_rank = np.empty(dense.shape[0], dtype=np.int64)
        cell_rank[np.asarray(adata.uns[key]["leaves"])] = np.arange(dense.shape[0])
    return _draw(dense, adata.obs[groupby], chr_names, chr_pos, groupby=groupby, cmap=cmap, figsize=figsize,
                 vmin=vmin, vmax=vmax, show=bool(show), save=save, ymin_lines=0,
                 dendrogram=bool(kwargs.pop("dendrogram", False)), cell_rank=cell_rank, **kwargs)


def chromosome_heatmap_summary(adata, *, groupby: str = "cnv_leiden", use_rep: str = "cnv", cmap="bwr",
                               figsize=(16, 10), show=None, save=None, **kwargs):
    """Heatmap of per-group mean smoothed expression, each group drawn 10 rows high (reference :95-193)."""
    if groupby == "cnv_leiden" and "cnv_leiden" not in adata.obs.columns:
        raise ValueError("'cnv_leiden' is not in `adata.obs`. Did you run `tl.leiden()`?")
    x = _host_matrix(adata.obsm[f"X_{use_rep}"])
    groups = _categories(adata.obs[groupby])  # reference: adata.obs[groupby].unique() of a categorical column
    labels = np.asarray(adata.obs[groupby].values)

    def group_mean(g):
        m = np.asarray(np.mean(x[labels == g, :], axis=0))
        return m.reshape(1, -1)

    mat = np.vstack([np.repeat(group_mean(g), 10, axis=0) for g in groups])
    rep_labels = np.hstack([np.repeat(g, 10) for g in groups])
    chr_names, chr_pos = _sorted_chr_pos(adata, use_rep)
    vmin = kwargs.pop("vmin", None)
    vmax = kwargs.pop("vmax", None)
    if vmin is None:
        vmin = np.min(mat)
    if vmax is None:
        vmax = np.max(mat)
    import pandas as pd

    rep_labels = pd.Categorical(rep_labels, categories=groups)  # keep the category order of the column
    return _draw(mat, rep_labels, chr_names, chr_pos, groupby=groupby, cmap=cmap, figsize=figsize, vmin=vmin,
                 vmax=vmax, show=bool(show), save=save, ymin_lines=-1,
                 dendrogram=bool(kwargs.pop("dendrogram", False)), **kwargs)

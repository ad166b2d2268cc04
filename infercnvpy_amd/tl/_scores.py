"""``tl.cnv_score`` -- drop-in for ``infercnvpy.tl.cnv_score`` (reference ``tl/_scores.py:14-74``)."""
from __future__ import annotations

import warnings

import numpy as np
import scipy.sparse as sp

from .. import _engine


def cnv_score(adata, groupby: str = "cnv_leiden", *, use_rep: str = "cnv", key_added: str = "cnv_score",
              inplace: bool = True, obs_key=None):
    """score[group] = mean(|X_cnv[cells of group, :]|), zeros included (reference :65-68).

    The per-cell ``sum |x|`` and the per-group sums run on the GPU in float64.  ``obsm["X_cnv"]`` may be the host CSR
    matrix of a host-input ``tl.infercnv`` call, a dense array, or the device-resident
    :class:`infercnvpy_amd.PackedCsr` of a call on an HBM-resident matrix (nothing is copied then).
    """
    if obs_key is not None:
        warnings.warn(
            "The obs_key argument has been renamed to `groupby` for consistency with "
            "other functions and will be removed in the future. ",
            category=FutureWarning,
            stacklevel=2,
        )
        groupby = obs_key
    if groupby not in adata.obs.columns and groupby == "cnv_leiden":
        raise ValueError("`cnv_leiden` not found in `adata.obs`. Did you run `tl.leiden`?")

    torch = _engine._torch()
    x = adata.obsm[f"X_{use_rep}"]
    n_win = x.shape[1]
    if isinstance(x, _engine.PackedCsr):  # X_cnv of a device-resident call: read where it lies, nothing is uploaded
        row_abs = _engine.csr_row_abs_sum(x)
    elif sp.issparse(x):  # what tl.infercnv writes: CSR float64 -- only the stored values travel, in float64
        row_abs = _engine.csr_row_abs_sum(x)
    else:
        x = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
        row_abs = _engine.row_abs_sum(torch.from_numpy(x).cuda())

    # group sums on the device in a fixed order (icv_group_sums): K sums come back, and a host X_cnv and a
    # device-resident one give the same bits.  Labels -> group codes in one vectorised pass (first-appearance order =
    # the reference's `labels.unique()` loop, :65); the counts are the host's own bincount
    import pandas as pd

    labels = adata.obs[groupby]
    if isinstance(getattr(labels, "dtype", None), pd.CategoricalDtype):
        # what tl.leiden leaves: the codes are there already (-1: missing); unused categories are groups without rows
        codes, uniques = labels.cat.codes.to_numpy(), labels.cat.categories
    else:
        codes, uniques = pd.factorize(np.asarray(labels.values if hasattr(labels, "values") else labels),
                                      use_na_sentinel=True)
    codes = codes.astype(np.int32, copy=False)
    n_groups = len(uniques)
    sums, _ = _engine.group_sums(row_abs, codes, n_groups, want_counts=False)
    counts = np.bincount(codes[codes >= 0], minlength=max(n_groups, 1))[:n_groups]
    with np.errstate(invalid="ignore", divide="ignore"):
        scores = sums / (counts.astype(np.float64) * n_win)
    pos = {u: gi for gi, u in enumerate(uniques)}
    # (a missing label is a "cluster" of the reference's loop too: no row equals it, its score is nan)
    cluster_score = {c: (np.float64(scores[pos[c]]) if c in pos else np.float64("nan")) for c in labels.unique()}

    if inplace:
        adata.obs[key_added] = np.where(codes >= 0, scores[np.maximum(codes, 0)] if n_groups else np.nan, np.nan)
    else:
        return cluster_score


def _group_iqr(adata, groupby, get_matrix, key_added, inplace):
    """Shared driver of ithcna / ithgex (reference tl/_scores.py:128-151, :197-221)."""
    torch = _engine._torch()
    labels = adata.obs[groupby]
    values = np.asarray(labels.values if hasattr(labels, "values") else labels)
    groups = labels.unique()
    scores = {}
    for group in groups:
        sel = values == group
        X = get_matrix(sel)
        if torch.is_tensor(X):  # rows of a device-resident PackedCsr, densified on the GPU (float32 tile)
            xd = X
        else:
            if sp.issparse(X):
                X = X.toarray()
            X = np.asarray(X)
            xd = None
        if X.shape[0] <= 1:
            continue
        if xd is None:
            xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).cuda()
        scores[group] = np.float64(_engine.corr_iqr(xd))
    if inplace:
        obs = np.empty(adata.shape[0])
        for group in groups:
            obs[values == group] = scores[group]  # KeyError for single-cell groups, as the reference
        adata.obs[key_added] = obs
    else:
        return scores


def ithgex(adata, groupby: str, *, use_raw=None, layer=None, inplace: bool = True, key_added: str = "ithgex"):
    """ITHGEX diversity score: IQR of the cell-cell Pearson correlations of gene expression per group.

    Drop-in for ``infercnvpy.tl.ithgex`` (reference tl/_scores.py:77-151); the correlation matrix is an
    fp32 MFMA contraction on the GPU.
    """
    if use_raw and layer is not None:
        raise ValueError(
            f"Cannot use expression from both layer and raw. You provided:'use_raw={use_raw}' and 'layer={layer}'")

    def get(sel):
        if layer is not None:
            return adata.layers[layer][sel]
        if use_raw:
            return adata.raw.X[sel]
        return adata.X[sel]

    return _group_iqr(adata, groupby, get, key_added, inplace)


def ithcna(adata, groupby: str, *, use_rep: str = "X_cnv", key_added: str = "ithcna", inplace: bool = True):
    """ITHCNA diversity score: IQR of the cell-cell Pearson correlations of the CNV profiles per group.

    Drop-in for ``infercnvpy.tl.ithcna`` (reference tl/_scores.py:154-221).
    """
    def get(sel):
        x = adata.obsm[use_rep]
        # X_cnv left on the device by tl.infercnv: the group's rows become a float32 tile without leaving HBM
        return x.dense_rows(sel) if isinstance(x, _engine.PackedCsr) else x[sel]

    return _group_iqr(adata, groupby, get, key_added, inplace)

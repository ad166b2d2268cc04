"""``tl.cnv_score`` -- drop-in for ``infercnvpy.tl.cnv_score`` (reference ``tl/_scores.py:14-74``)."""
from __future__ import annotations

import warnings

import numpy as np
import scipy.sparse as sp

from .. import _engine


def cnv_score(adata, groupby: str = "cnv_leiden", *, use_rep: str = "cnv", key_added: str = "cnv_score",
              inplace: bool = True, obs_key=None):
    """score[group] = mean(|X_cnv[cells of group, :]|), zeros included (reference :65-68).

    The per-cell ``sum |x|`` runs on the GPU in float64; groups are combined on the host.
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
    if sp.issparse(x):
        x = x.toarray()
    x = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
    row_abs = _engine.row_abs_sum(torch.from_numpy(x).cuda()).cpu().numpy()
    n_win = x.shape[1]

    labels = adata.obs[groupby]
    values = np.asarray(labels.values if hasattr(labels, "values") else labels)
    cluster_score = {}
    for cluster in labels.unique():
        sel = values == cluster
        cluster_score[cluster] = np.float64(row_abs[sel].sum() / (int(sel.sum()) * n_win))

    if inplace:
        adata.obs[key_added] = np.array([cluster_score[c] for c in adata.obs[groupby]])
    else:
        return cluster_score

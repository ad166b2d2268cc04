"""Cell-level Ward linkage of the CNV profiles (BASELINE.json config 5).

The reference draws its heatmap dendrogram through scanpy (``sc.pl.heatmap(..., dendrogram=True)``
forwarded from ``pl/_chromosome_heatmap.py:74-85``); it has no cell-level clustering of its own.
This module is the build-defined counterpart named by the north star: squared Euclidean distances
between all cells' ``X_cnv`` rows on fp32 MFMA tiles and Ward linkage on the resident matrix, both
on the GPU.  The result is a scipy linkage matrix (the oracle is ``scipy.spatial.distance.pdist`` +
``scipy.cluster.hierarchy.linkage(method="ward")``).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .. import _engine


def ward_linkage(X, *, return_rounds: bool = False):
    """Ward linkage (scipy format, ``(n - 1) x 4`` float64) of the rows of ``X`` (dense or sparse host matrix, or the
    device-resident :class:`infercnvpy_amd.PackedCsr` that ``tl.infercnv`` leaves in ``obsm["X_cnv"]`` for a matrix in
    HBM: config 5's input then never leaves the GPU).

    The distance matrix lives in HBM: ``6 n^2`` bytes with the spare columns the Ward rounds like (200 000 cells:
    240 GB of the 288 GB; Ward 0.53 s), ``4 n^2`` without them when memory is short (160 GB; Ward 0.97 s).  For
    more cells than one GPU holds: ``infercnvpy_amd.dist.ward_linkage_sharded``.
    """
    torch = _engine._torch()
    if isinstance(X, _engine.PackedCsr):  # X_cnv of a device-resident tl.infercnv call: densified in HBM
        if X.n_rows < 2:
            raise ValueError("at least two cells are needed for a linkage")
        xd = X.dense_rows()
        if not bool(torch.isfinite(xd).all()):
            raise ValueError("The condensed distance matrix must contain only finite values.")  # scipy's error
    else:
        if torch.is_tensor(X):
            X = X.detach().cpu().numpy()
        if sp.issparse(X):
            X = X.toarray()
        X = np.ascontiguousarray(np.asarray(X), dtype=np.float32)
        if X.ndim != 2:
            raise ValueError("X must be a 2-D matrix (cells x features)")
        if not np.isfinite(X).all():
            raise ValueError("The condensed distance matrix must contain only finite values.")  # scipy's error
        if X.shape[0] < 2:
            raise ValueError("at least two cells are needed for a linkage")
        xd = torch.from_numpy(X).cuda()
    d2 = _engine.pairwise_sqeuclidean(xd, spare=True)
    del xd
    Z, rounds = _engine.ward_linkage(d2, spare=_engine.has_spare_columns(d2))
    return (Z, rounds) if return_rounds else Z


def leaves_list(Z):
    """Left-to-right leaf order of a linkage matrix (iterative; scipy's ``leaves_list`` semantics)."""
    Z = np.asarray(Z)
    n = Z.shape[0] + 1
    left = Z[:, 0].astype(np.int64)
    right = Z[:, 1].astype(np.int64)
    out = np.empty(n, dtype=np.int64)
    k = 0
    stack = [2 * n - 2]
    while stack:
        node = stack.pop()
        if node < n:
            out[k] = node
            k += 1
        else:
            stack.append(right[node - n])
            stack.append(left[node - n])
    return out


def cell_linkage(adata, *, use_rep: str = "cnv", key_added: str | None = None, inplace: bool = True):
    """Ward linkage of all cells on ``adata.obsm["X_{use_rep}"]``.

    Stores ``{"linkage": Z, "leaves": order, "use_rep": use_rep}`` in ``adata.uns[key_added]``
    (default ``"{use_rep}_linkage"``) or returns ``Z``.
    """
    if f"X_{use_rep}" not in adata.obsm:
        raise KeyError(f"X_{use_rep} not found in adata.obsm. Did you run `tl.infercnv`?")
    Z = ward_linkage(adata.obsm[f"X_{use_rep}"])
    if not inplace:
        return Z
    adata.uns[key_added or f"{use_rep}_linkage"] = {"linkage": Z, "leaves": leaves_list(Z), "use_rep": use_rep}

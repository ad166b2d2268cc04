"""CPU oracle for the ``tl.infercnv`` hot path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithm implemented by the reference
(icbi-lab/infercnvpy, ``src/infercnvpy/tl/_infercnv.py`` and
``src/infercnvpy/tl/_scores.py``).  It is *not* part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported CPU baseline.
The product path (``infercnvpy_amd``) never imports anything from ``oracle/``
and fails loudly when the HIP extension is missing.

Parity pinning: ``tests/test_oracle_golden.py`` checks this file against
(1) the reference's own known-answer tests (transcribed arrays from
``tests/test_tools.py:64-191``, ``tests/conftest.py:61-108``,
``tests/test_scores.py:18-21``) and (2) vectors captured in the build container
by importing the reference's module (``tests/golden/make_golden.py``).  The
comparison is ``array_equal`` (bit-exact float64) for X_cnv and chr_pos.

Every function cites the reference lines (``file:line``, relative to the
reference root) whose behaviour it restates.  The arithmetic deliberately goes
through the same numpy primitives in the same dtype flow (``np.convolve`` of a
float32/float64 row with an int64 kernel -> float64; ``np.median``;
population ``np.std``) so the result is bit-identical, not merely close.
"""
from __future__ import annotations

import re
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import scipy.sparse as sp

__all__ = [
    "natural_order",
    "chromosome_gene_order",
    "window_weights",
    "smooth_segment",
    "smooth_all_chromosomes",
    "center_on_reference",
    "reference_profile",
    "infercnv_chunk",
    "infercnv",
    "gene_values_from_windows",
    "cnv_score",
    "ith_score",
    "ward_linkage",
]


# --------------------------------------------------------------------------- #
# ordering / indexing contract
# --------------------------------------------------------------------------- #
def natural_order(names):
    """chr1, chr2, ..., chr10, ... : digit runs compare as ints, text lower-cased.

    Restates ``_natural_sort`` (tl/_infercnv.py:164-176).
    """

    def key(name):
        parts = re.split("([0-9]+)", name)
        return [int(p) if p.isdigit() else p.lower() for p in parts]

    return sorted(names, key=key)


def _isnull(arr):
    arr = np.asarray(arr, dtype=object) if np.asarray(arr).dtype.kind in "OUS" else np.asarray(arr)
    if arr.dtype == object:
        return np.array([(v is None) or (isinstance(v, float) and v != v) for v in arr], dtype=bool)
    if arr.dtype.kind == "f":
        return np.isnan(arr)
    return np.zeros(arr.shape, dtype=bool)


def used_chromosomes(chrom):
    """Chromosomes that get windows: start with "chr", not "chrM", natural order.

    Restates tl/_infercnv.py:327 (other contigs are silently dropped).
    """
    seen = []
    for c in chrom:
        if c not in seen:
            seen.append(c)
    return natural_order([c for c in seen if isinstance(c, str) and c.startswith("chr") and c != "chrM"])


def chromosome_gene_order(chrom, start, name):
    """Column indices of chromosome ``name`` sorted by ``start``.

    Restates tl/_infercnv.py:350-351: ``var.loc[chromosome == chr]
    .sort_values("start")`` -> pandas ``nargsort`` = numpy quicksort argsort on the
    non-NaN starts, NaN starts appended last in original order.
    """
    idx = np.flatnonzero(np.asarray(chrom, dtype=object) == name)
    s = np.asarray(start)[idx]
    if s.dtype.kind == "f":
        nan = np.isnan(s)
        keep = idx[~nan]
        return np.concatenate([keep[np.argsort(s[~nan], kind="quicksort")], idx[nan]])
    return idx[np.argsort(s, kind="quicksort")]


# --------------------------------------------------------------------------- #
# smoothing
# --------------------------------------------------------------------------- #
def window_weights(n):
    """Pyramid weights min(r, reversed r), r = 1..n (tl/_infercnv.py:206-207)."""
    r = np.arange(1, n + 1)
    return np.minimum(r, r[::-1])


def smooth_segment(x, n, step):
    """Weighted running mean over one chromosome's genes (columns of ``x``).

    Restates ``_running_mean`` (tl/_infercnv.py:179-244):
    * ``n < G_c``: per-row ``np.convolve(row, pyramid, "valid") / sum(pyramid)``,
      then keep windows 0, step, 2*step, ... (:205-218);
    * otherwise one flat window = plain mean over all G_c genes (:227-236).
    """
    g = x.shape[1]
    if n < g:
        w = window_weights(n)
        full = np.stack([np.convolve(row, w, mode="valid") for row in x]) / np.sum(w)
        return full[:, np.arange(0, full.shape[1], step)]
    w = np.array([1] * g)
    return np.stack([np.convolve(row, w, mode="valid") for row in x]) / np.sum(w)


def smooth_all_chromosomes(x, chrom, start, n, step):
    """Per-chromosome smoothing, concatenated in natural chromosome order.

    Restates ``_running_mean_by_chromosome`` / ``_running_mean_for_chromosome``
    (tl/_infercnv.py:301-356).  Returns (chr_pos dict, C x W float64).
    """
    names = used_chromosomes(chrom)
    pieces = []
    for c in names:
        cols = chromosome_gene_order(chrom, start, c)
        pieces.append(smooth_segment(x[:, cols], n, step))
    chr_pos = {}
    offs = np.cumsum([0] + [p.shape[1] for p in pieces])
    for c, o in zip(names, offs):
        chr_pos[c] = o
    return chr_pos, np.hstack(pieces)


# --------------------------------------------------------------------------- #
# centring on the reference
# --------------------------------------------------------------------------- #
def _plain(a):
    """np.matrix -> ndarray (``_ensure_array``, _util.py:4-9)."""
    return a.A if isinstance(a, np.matrix) else a


def center_on_reference(x, reference):
    """Step 1 of the chunk kernel (tl/_infercnv.py:422-434).

    One reference row: plain subtraction (sparse input densifies).  Several rows:
    "bounded" difference -- value minus the per-gene max if above it, minus the
    per-gene min if below it, else 0; the result keeps ``x``'s dtype (:428).
    """
    if reference.shape[0] == 1:
        out = x - reference[0, :]
    else:
        lo = np.min(reference, axis=0)
        hi = np.max(reference, axis=0)
        out = np.zeros(x.shape, dtype=x.dtype)
        above = x > hi
        below = x < lo
        out[above] = _plain(x - hi)[above]
        out[below] = _plain(x - lo)[below]
    return _plain(out)


def reference_profile(X, obs_col, reference_cat, reference, n_vars):
    """R x G reference expression (``_get_reference``, tl/_infercnv.py:359-408).

    ``reference`` wins; else mean of all cells when key or cat is missing; else one
    mean per category (ValueError if a category is absent).  1-D -> 2-D; gene
    count is validated.
    """
    if reference is None:
        if obs_col is None or reference_cat is None:
            reference = np.mean(X, axis=0)
        else:
            if isinstance(reference_cat, str):
                reference_cat = [reference_cat]
            reference_cat = np.array(reference_cat)
            found = np.isin(reference_cat, obs_col)
            if not np.all(found):
                raise ValueError(f"reference categories not found: {reference_cat[~found]}")
            obs_vals = np.asarray(obs_col)
            reference = np.vstack([np.mean(X[obs_vals == cat, :], axis=0) for cat in reference_cat])
    if reference.ndim == 1:
        reference = reference[np.newaxis, :]
    if reference.shape[1] != n_vars:
        raise ValueError("Reference must match the number of genes.")
    return reference


# --------------------------------------------------------------------------- #
# per-gene values (calculate_gene_values=True)
# --------------------------------------------------------------------------- #
def gene_values_from_windows(smoothed, g, n, step):
    """Per-gene value = mean of the kept windows that contain the gene.

    Restates ``_calculate_gene_averages`` + ``get_convolution_indices``
    (tl/_infercnv.py:247-298) for one chromosome with ``g`` sorted genes:
    kept window ``j`` covers sorted genes ``[j*step, j*step + n)``.  Genes covered
    by no kept window are reported as NaN columns (reference: absent columns,
    NaN after reindex, :147).  Small chromosomes (``g <= n``): every gene gets the
    single window value (:238-240).
    """
    c = smoothed.shape[0]
    out = np.full((c, g), np.nan)
    if n < g:
        for p in range(g):
            js = [j for j in range(smoothed.shape[1]) if j * step <= p < j * step + n]
            if js:
                # np.mean over a python list of float64 scalars, in window order
                for i in range(c):
                    out[i, p] = np.mean([smoothed[i, j] for j in js])
    else:
        out[:, :] = smoothed[:, :1]
    return out


# --------------------------------------------------------------------------- #
# chunk kernel and driver
# --------------------------------------------------------------------------- #
def infercnv_chunk(x, chrom, start, reference, lfc_cap, window_size, step, dynamic_threshold,
                   calculate_gene_values=False):
    """Steps 1-5 for one chunk of cells (``_infercnv_chunk``, tl/_infercnv.py:411-457).

    Returns (chr_pos, dense float64 C x W, gene_res | None, noise_threshold | None).
    ``gene_res`` (if requested) is C x G_used in chromosome-sorted gene order with NaN
    for uncovered genes, plus the list of column indices it refers to.
    """
    centered = center_on_reference(x, reference)
    clipped = np.clip(centered, -lfc_cap, lfc_cap)
    chr_pos, smoothed = smooth_all_chromosomes(clipped, chrom, start, window_size, step)
    res = smoothed - np.median(smoothed, axis=1)[:, np.newaxis]

    gene_res = None
    if calculate_gene_values:
        cols, vals = [], []
        for c in used_chromosomes(chrom):
            order = chromosome_gene_order(chrom, start, c)
            seg = smooth_segment(clipped[:, order], window_size, step)
            gv = gene_values_from_windows(seg, len(order), window_size, step)
            covered = ~np.isnan(gv[0]) if gv.shape[0] else np.ones(len(order), bool)
            cols.append(order[covered])
            vals.append(gv[:, covered])
        cols = np.concatenate(cols)
        vals = np.hstack(vals)
        vals = vals - np.median(vals, axis=1)[:, np.newaxis]
        gene_res = (cols, vals)

    thr = None
    if dynamic_threshold is not None:
        thr = dynamic_threshold * np.std(res)
        res[np.abs(res) < thr] = 0
        if gene_res is not None:
            gene_res[1][np.abs(gene_res[1]) < thr] = 0
    return chr_pos, res, gene_res, thr


def _chunk_task(args):
    return infercnv_chunk(*args)


def infercnv(X, chrom, start, *, obs_col=None, reference_cat=None, reference=None, lfc_clip=3,
             window_size=100, step=10, dynamic_threshold=1.5, exclude_chromosomes=("chrX", "chrY"),
             chunksize=5000, n_jobs=1, calculate_gene_values=False):
    """Array-level restatement of the ``infercnv`` driver (tl/_infercnv.py:97-151).

    ``X``: C x G_all dense ndarray or scipy sparse; ``chrom``/``start``: length-G_all
    gene annotation.  Returns (chr_pos, csr float64 C x W, per_gene C x G_all | None,
    list of per-chunk thresholds).
    """
    chrom = np.asarray(chrom, dtype=object)
    drop = _isnull(chrom)
    if exclude_chromosomes is not None:
        drop = drop | np.isin(chrom, list(exclude_chromosomes))
    keep = ~drop
    ref = reference_profile(X, obs_col, reference_cat, reference, X.shape[1])[:, keep]
    expr = X[:, keep]
    if sp.issparse(expr):
        expr = expr.tocsr()
    ch, st = chrom[keep], np.asarray(start)[keep]

    tasks = [
        (expr[i: i + chunksize, :], ch, st, ref, lfc_clip, window_size, step, dynamic_threshold,
         calculate_gene_values)
        for i in range(0, X.shape[0], chunksize)
    ]
    if n_jobs and n_jobs > 1 and len(tasks) > 1:
        with ProcessPoolExecutor(max_workers=n_jobs) as pool:
            results = list(pool.map(_chunk_task, tasks))
    else:
        results = [_chunk_task(t) for t in tasks]

    chr_pos = results[0][0]
    res = sp.vstack([sp.csr_matrix(r[1]) for r in results])
    thrs = [r[3] for r in results]
    per_gene = None
    if calculate_gene_values:
        per_gene = np.full((X.shape[0], X.shape[1]), np.nan)
        kept_idx = np.flatnonzero(keep)
        row = 0
        for r in results:
            cols, vals = r[2]
            per_gene[row: row + vals.shape[0], kept_idx[cols]] = vals
            row += vals.shape[0]
    return chr_pos, res, per_gene, thrs


# --------------------------------------------------------------------------- #
# cnv_score
# --------------------------------------------------------------------------- #
def cnv_score(x_cnv, groups):
    """score[g] = mean(|X_cnv[rows of g, :]|) over all entries, zeros included.

    Restates ``cnv_score`` arithmetic (tl/_scores.py:65-68).  Group order = order
    of first appearance (``Series.unique``).
    """
    groups = np.asarray(groups, dtype=object)
    order = []
    for g in groups:
        if g not in order:
            order.append(g)
    return {g: np.mean(np.abs(x_cnv[groups == g, :])) for g in order}


def ith_score(X, groups):
    """IQR of all entries of the cell x cell Pearson correlation matrix, per group.

    Restates ``ithgex`` / ``ithcna`` (tl/_scores.py:128-144, :197-213): groups with <= 1 cell are skipped.
    """
    groups = np.asarray(groups, dtype=object)
    order = []
    for g in groups:
        if g not in order:
            order.append(g)
    out = {}
    for g in order:
        x = X[groups == g]
        if sp.issparse(x):
            x = x.todense()
        if x.shape[0] <= 1:
            continue
        pcorr = np.corrcoef(x, rowvar=True)
        q75, q25 = np.percentile(pcorr, [75, 25])
        out[g] = q75 - q25
    return out


def ward_linkage(X):
    """Build-defined oracle of BASELINE config 5 (SURVEY.md 8(c) iii): the reference has no call site for
    cell-level clustering, so parity is anchored on scipy (1.15.3 here): pdist (float64 Euclidean) +
    linkage(method="ward").  PARITY UNPINNED against the reference."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist

    if sp.issparse(X):
        X = X.toarray()
    return linkage(pdist(np.asarray(X, dtype=np.float64)), method="ward")

"""``tl.infercnv`` -- drop-in for ``infercnvpy.tl.infercnv`` on AMD MI355X.

Same keyword-only signature, defaults, error behaviour and AnnData side effects as the reference
driver (icbi-lab/infercnvpy ``src/infercnvpy/tl/_infercnv.py:18-161``).  The numeric work of the
reference's ``_infercnv_chunk`` (:411-457) and ``_get_reference`` (:359-408) runs in hand-written
gfx950 kernels behind the C ABI ``include/infercnv_hip.h``; this module only validates, plans the
gene order, moves data and writes the result fields.

Parallelism lives inside the call, as in the reference (``process_map`` over row chunks driven by
``n_jobs``, :28, :120-135): the rows are cut into contiguous, ``chunksize``-aligned shards, one per
GPU (``n_jobs`` / ``devices``), every shard is streamed from host memory to its own GPU by its own
uploader, smoothed there, packed to CSR there and copied back; the host concatenates the shards in
row order (``vstack``, :137).
"""
from __future__ import annotations

import logging
import os
import sys
import threading
import time as _time
from collections.abc import Sequence

import numpy as np
import scipy.sparse as sp

from .. import _engine, _lib
from .._plan import GenePlan

log = logging.getLogger("infercnvpy_amd")

# n_jobs=None (the reference's "all cores"): an extra GPU is used only if every GPU gets at least this many chunks and
# this many cells (a GPU smooths 30 M cells/s: small inputs are not worth a second context and uploader)
_MIN_CHUNKS_PER_DEVICE = 4
_MIN_ROWS_PER_DEVICE = 50_000


def _reference_groups(obs, reference_key, reference_cat):
    """Row -> group index (-1: none), per-category counts and the category list for per-category means
    (reference ``_get_reference``, :388-400).  Raises the reference's ValueError for absent categories."""
    obs_col = obs[reference_key]
    if isinstance(reference_cat, str):
        reference_cat = [reference_cat]
    cats = np.array(reference_cat)
    present = np.isin(cats, obs_col)
    if not np.all(present):
        raise ValueError(
            f"The following reference categories were not found in adata.obs[reference_key]: {cats[~present]}")
    obs_vals = np.asarray(obs_col.values if hasattr(obs_col, "values") else obs_col)
    groups = np.full(len(obs_vals), -1, dtype=np.int32)
    counts = np.zeros(len(cats), dtype=np.int64)
    # a cell belongs to the first listed category it equals (categories are distinct labels)
    for gi, cat in enumerate(cats):
        sel = obs_vals == cat
        counts[gi] = int(sel.sum())
        groups[sel & (groups < 0)] = gi
    return groups, counts, cats


def _means_from_chains(accs, counts, cats, is_sparse):
    """R x G means from the reference-order accumulators of the LAST shard (one array of the matrix dtype per group).
    Dense input: numpy's own last step, ``true_divide(sum, n)`` in that dtype; CSR input: scipy scaled every entry by
    1 / n before adding, the accumulators are the means."""
    rows = []
    for gi, acc in enumerate(accs):
        n = int(counts if cats is None else counts[gi])
        rows.append(acc if is_sparse else np.true_divide(acc, n))
    if cats is not None:
        labels = cats.tolist()
        if len(set(labels)) != len(labels):  # same label listed twice: rows repeat
            first = {c: i for i, c in reversed(list(enumerate(labels)))}
            rows = [rows[first[c]] for c in labels]
    return np.vstack(rows)


def _means_from_sums(shard_sums, counts, cats, dtype):
    """R x G means from the shards' float64 column sums (``mean_order="float64"``): the shards are added in row order,
    the quotient is rounded once to the matrix dtype."""
    total = shard_sums[0].copy()
    for part in shard_sums[1:]:
        total += part
    cnt = np.atleast_1d(np.asarray(counts, dtype=np.float64))
    rows = list((total / cnt[:, None]).astype(dtype))
    return np.vstack(_repeat_rows(rows, cats))


def _current_first(torch, n_dev):
    """The visible GPUs, the caller's current device first (a caller that did ``torch.cuda.set_device(3)`` gets GPU 3
    for ``n_jobs=1`` and GPU 3 among the first for more: ADVICE r3)."""
    cur = torch.cuda.current_device()
    return [cur] + [d for d in range(n_dev) if d != cur]


def _resolve_devices(n_jobs, devices, n_chunks, torch, n_obs=None):
    """GPU of every row shard.  ``devices`` wins (a device may be listed more than once: several shards share it);
    else ``n_jobs`` GPUs, the current device first; else all visible GPUs (current first), as far as each gets a few
    chunks of work.  Never more shards than chunks."""
    n_dev = torch.cuda.device_count()
    if devices is not None:
        devs = [torch.device(d).index if not isinstance(d, (int, np.integer)) else int(d) for d in devices]
        devs = [torch.cuda.current_device() if d is None else d for d in devs]
        if not devs:
            raise ValueError("devices must name at least one GPU")
        for d in devs:
            if not 0 <= d < n_dev:
                raise ValueError(f"devices: GPU {d} requested, {n_dev} visible")
    elif n_jobs is not None and int(n_jobs) >= 1:
        k = min(int(n_jobs), n_dev)
        devs = _current_first(torch, n_dev)[:k]
    else:
        in_group = False
        try:
            import torch.distributed as td

            in_group = td.is_available() and td.is_initialized() and td.get_world_size() > 1
        except Exception:
            in_group = False
        # inside a torch.distributed job every rank owns ONE GPU: never reach for the others
        k = 1 if in_group else max(1, min(n_dev, n_chunks // _MIN_CHUNKS_PER_DEVICE))
        if n_obs is not None:
            k = max(1, min(k, int(n_obs) // _MIN_ROWS_PER_DEVICE))
        devs = _current_first(torch, n_dev)[:k]
        if k > 1:
            log.info(f"tl.infercnv: {n_obs} cells over GPUs {devs} (n_jobs=None: every visible GPU with enough work)")
    return devs[: max(1, n_chunks)]


class _Shard:
    """The rows [g0, g1) of the matrix on one GPU: plan, slabs, upload streams, kernels, CSR drain."""

    def __init__(self, index, device, share, g0, g1):
        self.index, self.device, self.share = index, device, share
        self.g0, self.g1 = g0, g1
        self.tm = {}
        self.result = None      # (indptr, indices, data) of the shard's X_cnv
        self.gene_pieces = []
        self.accs = None        # host, per group: reference-order accumulators after this shard's rows


_PLAN_CACHE = {}  # key -> _PlanEntry, most recently used last
_PLAN_CACHE_LOCK = threading.Lock()
_PLAN_CACHE_MAX = 8


class _PlanEntry:
    """The plans of one (annotation, geometry, device): idle ones wait for the next call, ``busy`` counts the calls
    that hold one.  A plan admits one compute call at a time (it owns the per-call device workspace), so concurrent
    callers each check out their own."""

    def __init__(self, make):
        self.make, self.idle, self.busy, self.evicted = make, [], 0, False


def _plan_key(var_chrom, var_start, window_size, step, exclude_chromosomes, device):
    """Key by the CONTENT of the annotation: object / string columns through their factorised codes and category
    strings (not through object addresses: ADVICE r4), numeric columns through their bytes."""
    import pandas as pd

    chrom, start = np.asarray(var_chrom), np.asarray(var_start)
    if chrom.dtype.kind in "OUS":
        codes, uniques = pd.factorize(chrom, use_na_sentinel=True)
        ckey = (codes.astype(np.int32).tobytes(), tuple(str(u) for u in uniques))
    else:
        ckey = (chrom.dtype.str, chrom.tobytes())
    if start.dtype.kind == "O":
        skey = ("O", pd.to_numeric(pd.Series(start), errors="coerce").to_numpy(dtype=np.float64, na_value=np.nan).tobytes())
    else:
        skey = (start.dtype.str, start.tobytes())
    excl = None if exclude_chromosomes is None else tuple(exclude_chromosomes)
    return (ckey, skey, int(window_size), int(step), excl, int(device))


def _plan_entry(var_chrom, var_start, window_size, step, exclude_chromosomes, device):
    """(key, entry) of the cache, created and moved to the most-recent end; evicts the least recently used entries
    that no call holds (their idle plans are closed; an entry in use is closed when its last plan comes back)."""
    key = _plan_key(var_chrom, var_start, window_size, step, exclude_chromosomes, device)
    ent = _PLAN_CACHE.pop(key, None)
    if ent is None:
        chrom, start = np.array(var_chrom, copy=True), np.array(var_start, copy=True)  # (not a view of the DataFrame)

        def make():
            return GenePlan(chrom, start, window_size=window_size, step=step, exclude_chromosomes=exclude_chromosomes)

        ent = _PlanEntry(make)
    _PLAN_CACHE[key] = ent
    for k in list(_PLAN_CACHE):
        if len(_PLAN_CACHE) <= _PLAN_CACHE_MAX:
            break
        if k == key:
            continue
        old = _PLAN_CACHE.pop(k)
        old.evicted = True
        while old.idle:
            old.idle.pop().close()
    return key, ent


def _checkout_plan(var_chrom, var_start, window_size, step, exclude_chromosomes, device):
    """A GenePlan for ONE call on an HBM-resident matrix, kept between calls: planning the gene order costs ~9 ms of
    host time at 20 000 genes, three times the GPU time of 100 000 cells.  Returns ``(entry, plan)``; hand the plan
    back with :func:`_checkin_plan`.  Two threads calling at once get two plans (a shared plan answered the
    second with "plan busy", and an eviction could destroy a plan in use)."""
    with _PLAN_CACHE_LOCK:
        _, ent = _plan_entry(var_chrom, var_start, window_size, step, exclude_chromosomes, device)
        plan = ent.idle.pop() if ent.idle else None
        ent.busy += 1
    if plan is None:
        try:
            plan = ent.make()
        except BaseException:
            with _PLAN_CACHE_LOCK:
                ent.busy -= 1
            raise
    return ent, plan


def _checkin_plan(ent, plan):
    with _PLAN_CACHE_LOCK:
        ent.busy -= 1
        if ent.evicted:
            plan.close()
        else:
            ent.idle.append(plan)


def _cached_plan(var_chrom, var_start, window_size, step, exclude_chromosomes, device):
    """The idle plan the next single-threaded resident call with these arguments will use (created if there is none):
    for callers that attach to it between calls (``bench.py``: ``icv_profile_begin`` / ``_collect``)."""
    with _PLAN_CACHE_LOCK:
        _, ent = _plan_entry(var_chrom, var_start, window_size, step, exclude_chromosomes, device)
        if not ent.idle:
            ent.idle.append(ent.make())
        return ent.idle[-1]


def _clear_plan_cache():
    with _PLAN_CACHE_LOCK:
        while _PLAN_CACHE:
            ent = _PLAN_CACHE.popitem()[1]
            ent.evicted = True
            while ent.idle:
                ent.idle.pop().close()


import atexit as _atexit  # noqa: E402

_atexit.register(_clear_plan_cache)  # cached plans own device buffers: release them before the HIP runtime goes


def _resident_matrix(X, torch):
    """``adata.X`` that already lives in HBM: a CUDA ``torch.Tensor`` (cells x genes, float32 / float64) or an
    ``infercnvpy_amd.DeviceMatrix`` (dense or CSR device arrays); None for host data."""
    if isinstance(X, _engine.DeviceMatrix):
        return X
    if torch is not None and isinstance(X, torch.Tensor) and X.is_cuda:
        if X.dim() != 2 or X.dtype not in (torch.float32, torch.float64):
            raise ValueError("a device matrix must be a 2-D float32 / float64 tensor")
        return _engine.DeviceMatrix(dense=X if X.stride(1) == 1 else X.contiguous())
    return None


def _is_real_anndata(adata):
    mod = type(adata).__module__ or ""
    return mod == "anndata" or mod.startswith("anndata.")


def _repeat_rows(rows, cats):
    """Per-category rows in the order the categories were listed (the same label listed twice: its row repeats)."""
    if cats is None:
        return rows
    labels = cats.tolist()
    if len(set(labels)) == len(labels):
        return rows
    first = {c: i for i, c in reversed(list(enumerate(labels)))}
    return [rows[first[c]] for c in labels]


def _infercnv_resident(var, obs, dm, *, reference_key, reference_cat, reference, lfc_clip, window_size, step,
                       dynamic_threshold, exclude_chromosomes, chunksize, calculate_gene_values, mean_order, tm):
    """The call on a matrix that is resident in HBM (no host copies, no synchronisation): reference means as chains on
    the device, one smoothing launch per piece, threshold + CSR pack; returns ``(chr_pos, PackedCsr, gene values |
    None)`` with everything still on the device."""
    torch = _engine._torch()
    t_start = _time.perf_counter()
    dev = dm._keep[0].device.index
    n_obs, n_vars = dm.shape
    with torch.cuda.device(dev):
        ent, plan = _checkout_plan(var["chromosome"].to_numpy(), var["start"].to_numpy(), window_size, step,
                                   exclude_chromosomes, dev)
        try:
            if plan.n_without_position:
                log.warning(f"Skipped {plan.n_without_position} genes because they don't have a genomic position annotated. ")
            is_csr = dm.format == _lib.ICV_CSR
            np_dtype = np.float32 if dm.dtype == torch.float32 else np.float64
            flags = 0
            if reference is not None:
                given = np.asarray(reference)
                if given.ndim == 1:
                    given = given[np.newaxis, :]
                if given.shape[1] != n_vars:
                    raise ValueError("Reference must match the number of genes in AnnData. ")
                if np.result_type(np_dtype, given.dtype) != np_dtype:
                    raise ValueError("a device-resident matrix needs a reference of its own (or a narrower) dtype: "
                                     "numpy would promote the subtraction (pass the matrix as float64)")
                ref = torch.from_numpy(np.ascontiguousarray(given.astype(np_dtype))).cuda()
            else:
                groups = cats = None
                if reference_key is None or reference_cat is None:
                    log.warning("Using mean of all cells as reference. For better results, provide either "
                                "`reference`, or both `reference_key` and `reference_cat`. ")
                    counts = [n_obs]
                else:
                    groups, counts, cats = _reference_groups(obs, reference_key, reference_cat)
                if mean_order == "float64":
                    # opt-in: float64 column sums (one concurrent pass, correctly rounded means -- not numpy's order)
                    sums = _engine.column_sums(dm, groups, len(counts))
                    cnt = torch.as_tensor(np.asarray(counts, dtype=np.float64)).cuda()
                    rows = list((sums / cnt[:, None]).to(dm.dtype))
                else:
                    rows = []
                    for gi, n_g in enumerate(counts):
                        sel = None if groups is None else np.nonzero(groups == gi)[0]
                        acc = _engine.column_chain(dm, None, sel, int(n_g))
                        rows.append(_engine.chain_mean(acc, int(n_g), is_csr))
                ref = torch.stack(_repeat_rows(rows, cats))
            if ref.shape[0] == 1:
                ref_lo, ref_hi = ref[0].contiguous(), None
            else:
                ref_lo, ref_hi = ref.min(dim=0).values.contiguous(), ref.max(dim=0).values.contiguous()
            # pieces of whole chunks whose result buffers (4 + 12 bytes per window, worst case) fit next to the matrix
            free_b = _engine.free_hbm_bytes()
            per_row = 16 * plan.n_windows + 64 + (8 * (n_vars + plan.n_windows) if calculate_gene_values else 0)
            if calculate_gene_values:
                free_b = max(free_b - 8 * n_vars * n_obs, free_b // 8)  # the float64 gene matrix of ALL rows stays
            piece = max(chunksize, int(0.4 * free_b // per_row) // chunksize * chunksize)
            parts = []
            genes = torch.empty((n_obs, n_vars), dtype=torch.float64, device="cuda") if calculate_gene_values else None
            for r0 in range(0, max(n_obs, 1), piece):
                r1 = min(n_obs, r0 + piece)
                # (calculate_gene_values: the smoothing launch also writes its float64 windows -- one pass, as the
                # reference's _infercnv_chunk returns both from one pass, :438-457)
                res = _engine.run_hot_path(plan, dm, ref_lo, ref_hi, lfc_clip=lfc_clip,
                                           dynamic_threshold=dynamic_threshold, chunksize=chunksize, flags=flags,
                                           row0=r0, row1=r1, apply=False, windows=calculate_gene_values)
                parts.append(_engine.threshold_csr(plan, dm, ref_lo, ref_hi, res, lfc_clip=lfc_clip,
                                                   chunksize=chunksize, flags=flags, row0=r0, row1=r1))
                if calculate_gene_values and r1 > r0:
                    # (pieces are whole chunks: the piece's thresholds are the thresholds of its rows' chunks)
                    _engine.gene_values_from_windows(plan, res.windows, thr=res.thr, chunksize=chunksize,
                                                     out=genes[r0:r1])
                del res
            x_cnv = parts[0] if len(parts) == 1 else _engine.concat_packed(parts)
            tm["kernel"] = plan.last_kernel()
            tm["devices"] = [dev]
            tm["pieces"] = len(parts)
            tm["total"] = _time.perf_counter() - t_start  # host time only: nothing has been waited for
            chr_pos = dict(plan.chr_pos)
        finally:
            _checkin_plan(ent, plan)
    return chr_pos, x_cnv, genes


def _check_var(var, var_names):
    if not var_names.is_unique:
        raise ValueError("Ensure your var_names are unique!")
    if {"chromosome", "start", "end"} - set(var.columns) != set():
        raise ValueError(
            "Genomic positions not found. There need to be `chromosome`, `start`, and `end` columns in `adata.var`. ")


def _check_mean_order(mean_order):
    if mean_order not in ("reference", "float64"):
        raise ValueError("mean_order must be 'reference' (numpy's / scipy's evaluation order) or 'float64'")


def infercnv_device(
    X,
    var,
    obs=None,
    *,
    reference_key: str | None = None,
    reference_cat: None | str | Sequence[str] = None,
    reference: np.ndarray | None = None,
    lfc_clip: float = 3,
    window_size: int = 100,
    step: int = 10,
    dynamic_threshold: float | None = 1.5,
    exclude_chromosomes: Sequence[str] | None = ("chrX", "chrY"),
    chunksize: int = 5000,
    calculate_gene_values: bool = False,
    mean_order: str = "reference",
    _timings: dict | None = None,
):
    """``tl.infercnv`` for a matrix that already lives in HBM, without an AnnData container (not part of the reference
    API): ``X`` is a CUDA ``torch.Tensor`` (cells x genes, float32 / float64) or an ``infercnvpy_amd.DeviceMatrix``,
    ``var`` the genes' annotation (``adata.var``: ``chromosome``, ``start``, ``end``, unique index), ``obs`` the
    cells' (``adata.obs``; needed for ``reference_key``).  The other arguments mean what they mean in
    :func:`infercnv`.

    Returns ``(chr_pos, x_cnv, gene_values)``: ``x_cnv`` an ``infercnvpy_amd.PackedCsr`` (device CSR float64),
    ``gene_values`` a device float64 tensor or None.  Nothing is copied from or to the host and the call does not wait
    for the GPU.  This is the entry point for users of a real ``anndata.AnnData``, which does not accept device
    tensors in ``X`` / ``layers`` / ``obsm``::

        chr_pos, x_cnv, _ = cnv.tl.infercnv_device(X_gpu, adata.var, adata.obs, reference_key="cell_type", ...)
        adata.obsm["X_cnv"] = x_cnv.to_scipy()          # the reference's host matrix
        adata.uns["cnv"] = {"chr_pos": chr_pos}
    """
    _check_var(var, var.index)
    _check_mean_order(mean_order)
    _lib.load()
    dm = _resident_matrix(X, sys.modules.get("torch"))
    if dm is None:
        raise ValueError("infercnv_device: X must be a CUDA torch.Tensor or an infercnvpy_amd.DeviceMatrix "
                         "(host matrices go through tl.infercnv)")
    if dm.shape[1] != len(var):
        raise ValueError("X must have one column per row of `var`")
    if reference is None and reference_key is not None and reference_cat is not None and obs is None:
        raise ValueError("reference_key needs `obs`")
    chunksize = int(chunksize)
    if chunksize < 1:
        raise ValueError("chunksize must be >= 1")
    return _infercnv_resident(
        var, obs, dm, reference_key=reference_key, reference_cat=reference_cat, reference=reference,
        lfc_clip=lfc_clip, window_size=window_size, step=step, dynamic_threshold=dynamic_threshold,
        exclude_chromosomes=exclude_chromosomes, chunksize=chunksize, calculate_gene_values=calculate_gene_values,
        mean_order=mean_order, tm=_timings if _timings is not None else {})


def infercnv(
    adata,
    *,
    reference_key: str | None = None,
    reference_cat: None | str | Sequence[str] = None,
    reference: np.ndarray | None = None,
    lfc_clip: float = 3,
    window_size: int = 100,
    step: int = 10,
    dynamic_threshold: float | None = 1.5,
    exclude_chromosomes: Sequence[str] | None = ("chrX", "chrY"),
    chunksize: int = 5000,
    n_jobs: int | None = None,
    inplace: bool = True,
    layer: str | None = None,
    key_added: str = "cnv",
    calculate_gene_values: bool = False,
    devices: Sequence[int] | None = None,
    mean_order: str = "reference",
    _timings: dict | None = None,
):
    """Infer copy number variation by averaging gene expression over genomic regions (GPU).

    Parameters and return value as the reference function (``tl/_infercnv.py:18-96``).

    ``n_jobs`` keeps the reference's meaning -- how many workers share the row chunks (:28, :120-135) -- with GPUs
    as the workers: ``n_jobs=k`` shards the rows over the first ``k`` visible GPUs, ``n_jobs=1`` uses the current
    device only, ``None`` (the reference's "all cores") uses every visible GPU that would get at least
    four chunks and 50 000 cells (one GPU inside a ``torch.distributed`` job).  ``devices`` (not part of the reference API) names
    the GPUs explicitly; a GPU listed twice carries two shards.  Shard boundaries are multiples of ``chunksize``,
    so the noise threshold -- the standard deviation of each ``chunksize``-cell chunk, reference :449-451 -- never
    couples two shards and ``X_cnv`` does not depend on the number of GPUs.  When the reference profile is a mean
    over cells (``reference=None``), it is formed in the reference's own evaluation order -- numpy adds a C-contiguous
    matrix row by row in the matrix dtype, scipy adds ``x * (1 / n)`` row by row for CSR and reduces CSC columns
    pairwise (reference :385, :400) -- so the means, and with them ``X_cnv``, equal the reference's bit for bit.  A
    float32 chain is sequential per column: with several GPUs shard k continues the accumulators of shard k - 1 (R x G
    values travel through the host); the uploads of all shards still overlap.  A dense matrix stored column-major
    (``np.asfortranarray``, a transposed genes x cells array) gets numpy's order for THAT layout (pairwise per column
    over 8 192-element pieces), formed on the first GPU before the shards start.  Not reproduced: scipy's order for
    sparse formats other than CSR / CSC (converted to CSR; a warning says so), and the per-category means of a dense
    matrix with ONE column (``X[rows, :]`` is then a contiguous vector that numpy reduces pairwise; the all-cell mean of
    such a matrix does take that route).
    ``_timings`` (not part of the reference API): a dict that receives the wall-clock seconds of the stages (plan,
    host -> HBM copy, kernels, CSR pack + copy back; per shard under ``"shards"`` when there are several).

    Precision of ``X_cnv``: every window is accumulated, centred and compared with the noise threshold in float64
    (as the reference's ``np.convolve`` is) and stored on the device as float32; the CSR values are those float32
    numbers widened to float64.  For float64 / integer input the reference keeps full float64 values, so entries
    differ from it by up to half a float32 ulp (|x| <= lfc_clip = 3: 1.2e-7 absolute; the tests bound 1e-6).
    The zero pattern follows the float64 comparison: a float32 value that ties with the rounded threshold is
    re-decided from a float64 re-evaluation of its window.  Float64 evaluation orders differ between kernels (and
    from numpy's) at the 1e-12 level -- CSR input with long windows sums its windows from prefix sums -- so an
    entry within ~1e-12 of the threshold may fall on the other side than in the reference (none in the golden
    vectors; expected well below one entry per 10^9).

    A matrix that already lives in HBM -- ``adata.X`` (or the layer) a CUDA ``torch.Tensor`` or an
    ``infercnvpy_amd.DeviceMatrix`` (dense or CSR device arrays) -- is processed in place on its GPU: nothing is copied
    from or to the host, the call returns without waiting for the GPU, and ``obsm["X_cnv"]`` is an
    ``infercnvpy_amd.PackedCsr`` (device CSR; ``.to_scipy()`` gives the reference's host matrix, bit-identical to the
    host-input call).  ``n_jobs`` / ``devices`` do not apply to it.  ``tl.cnv_score``, ``tl.ithcna``,
    ``tl.cell_linkage`` and ``pl.chromosome_heatmap(_summary)`` take that object as they take the host matrix.  This
    works with the duck-typed ``infercnvpy_amd.SimpleAnnData`` (any container with ``X, obs, var, obsm, uns, layers``);
    a real ``anndata.AnnData`` refuses device tensors in ``X``, so its users call :func:`infercnv_device` (same
    arguments, explicit ``X, var, obs``) and store ``x_cnv.to_scipy()`` -- and should this function ever be handed a
    real AnnData with a resident matrix, it writes the reference's host objects into it.

    ``mean_order`` (not part of the reference API; applies when the reference profile is a mean over cells):
    ``"reference"`` (default) forms the means in numpy's / scipy's own evaluation order -- the reference's bits, but a
    sequential chain per column, so several GPUs take turns on it; ``"float64"`` is the opt-in for multi-GPU jobs that
    prefer speed: every GPU adds float64 column sums of its rows concurrently and the host adds the shards' sums
    (correctly rounded means; ~1e-3 of the ``X_cnv`` entries next to the noise threshold may fall on the other side
    than in the reference).

    Data movement: the rows are copied to HBM in pieces of a few chunks by a helper thread on a side stream
    while the pieces that have landed are smoothed (reference means: chained); a dense float matrix that is mostly
    zeros (fewer than 30 % non-zeros in a probe of its first rows -- log-counts usually are) crosses PCIe as its stored
    entries (8 bytes each, packed by host threads) and is rebuilt as the same dense rows in HBM; the noise threshold and the CSR
    packing of X_cnv run on the GPU from the un-thresholded result and a keep-mask (x_res is never
    rewritten) and only the packed arrays cross PCIe on the way back.
    """
    tm = _timings if _timings is not None else {}
    t_start = _time.perf_counter()
    _check_var(adata.var, adata.var_names)
    _check_mean_order(mean_order)
    _lib.load()  # fail loudly before doing any work if the HIP extension is missing

    X0 = adata.X if layer is None else adata.layers[layer]
    dm0 = _resident_matrix(X0, sys.modules.get("torch"))
    if dm0 is not None:
        chunksize = int(chunksize)
        if chunksize < 1:
            raise ValueError("chunksize must be >= 1")
        chr_pos, x_cnv, per_gene = _infercnv_resident(
            adata.var, adata.obs, dm0, reference_key=reference_key, reference_cat=reference_cat, reference=reference,
            lfc_clip=lfc_clip, window_size=window_size, step=step, dynamic_threshold=dynamic_threshold,
            exclude_chromosomes=exclude_chromosomes, chunksize=chunksize,
            calculate_gene_values=calculate_gene_values, mean_order=mean_order, tm=tm)
        if not inplace:
            return chr_pos, x_cnv, per_gene
        if _is_real_anndata(adata):
            # anndata validates what goes into obsm / layers and takes neither a PackedCsr nor a device tensor: a real
            # AnnData gets the reference's host objects (tl.infercnv_device keeps the result on the GPU)
            x_cnv = x_cnv.to_scipy()
            per_gene = per_gene.cpu().numpy() if per_gene is not None else None
        adata.obsm[f"X_{key_added}"] = x_cnv
        adata.uns[key_added] = {"chr_pos": chr_pos}
        if calculate_gene_values:
            adata.layers[f"gene_values_{key_added}"] = per_gene
        return None

    var_chrom, var_start = adata.var["chromosome"].to_numpy(), adata.var["start"].to_numpy()

    X = adata.X if layer is None else adata.layers[layer]
    if isinstance(X, np.matrix):
        X = np.asarray(X)
    X_csc = X if (sp.issparse(X) and X.format == "csc") else None  # scipy reduces CSC columns in another order
    # a dense matrix stored column-major (np.asfortranarray, the transposed view of a genes x cells array): numpy puts the
    # axis with the smaller stride innermost and reduces every column pairwise instead of as one chain (:385)
    # (a single-column matrix is a 1-D contiguous reduction for numpy: the same pairwise order)
    X_fortran = X if (isinstance(X, np.ndarray) and X.ndim == 2 and X.shape[0] > 1 and
                      (abs(X.strides[0]) < abs(X.strides[1]) or X.shape[1] == 1)) else None
    if sp.issparse(X) and X.format not in ("csr", "csc") and reference is None and mean_order == "reference":
        log.warning(f"tl.infercnv: a {X.format.upper()} matrix is converted to CSR; scipy's own summation order for this "
                    "format is not reproduced, so the reference means (and entries of X_cnv next to the noise "
                    "threshold) may differ from infercnvpy's in the last bit.  Convert with .tocsr() for exact parity.")
    if sp.issparse(X):
        X = X.tocsr()
        if not X.has_canonical_format:  # the kernels expect unique, sorted column indices per row
            X = X.copy()
            X.sum_duplicates()
    n_obs, n_vars = X.shape
    chunksize = int(chunksize)
    if chunksize < 1:
        raise ValueError("chunksize must be >= 1")

    torch = _engine._torch()

    # ---- compute dtype: what numpy's promotion gives the reference (:423, :428) ----------------
    x_is_int = X.dtype.kind in "iub"
    given = None
    if reference is not None:
        given = np.asarray(reference)
    mean_dtype = np.float32 if X.dtype == np.float32 else np.float64
    ref_dtype = given.dtype if given is not None else mean_dtype
    if x_is_int or X.dtype == np.float64:
        compute = np.float64
    elif X.dtype == np.float32 or X.dtype == np.float16:
        compute = np.float32 if np.result_type(np.float32, ref_dtype) == np.float32 else np.float64
    else:
        raise ValueError(f"unsupported matrix dtype {X.dtype}")
    tdtype = torch.float32 if compute == np.float32 else torch.float64
    esz = 4 if compute == np.float32 else 8

    # ---- the reference profile: given, or per-group column means formed from the shards' sums (:359-408) --------
    need_means = reference is None
    groups = counts = cats = None
    n_groups = 1
    if need_means:
        if reference_key is None or reference_cat is None:
            log.warning("Using mean of all cells as reference. For better results, provide either "
                        "`reference`, or both `reference_key` and `reference_cat`. ")
            counts = float(n_obs)
        else:
            groups, counts, cats = _reference_groups(adata.obs, reference_key, reference_cat)
            n_groups = len(cats)
    else:
        given = np.asarray(given)
        if given.ndim == 1:
            given = given[np.newaxis, :]
        if given.shape[1] != n_vars:
            raise ValueError("Reference must match the number of genes in AnnData. ")

    # ---- row shards: contiguous, chunk-aligned, one per listed GPU (reference: chunks to a process pool) ---------
    from ..dist import shard_bounds

    n_chunks = -(-n_obs // chunksize) if n_obs else 0
    devs = _resolve_devices(n_jobs, devices, n_chunks, torch, n_obs)
    bounds = [b for b in shard_bounds(n_obs, len(devs), chunksize) if b[1] > b[0]] or [(0, n_obs)]
    devs = devs[: len(bounds)]
    shards = [_Shard(i, d, devs.count(d), g0, g1) for i, (d, (g0, g1)) in enumerate(zip(devs, bounds))]
    multi = len(shards) > 1
    tm["devices"] = list(devs)

    f64_means = need_means and mean_order == "float64"
    if need_means and X_csc is not None and not f64_means:
        # CSC input: np.add.reduceat per column over ALL rows of the group -- not separable by row shards; the CSC
        # arrays go to the first GPU in column blocks (values + row indices, 8 bytes per stored entry)
        t0 = _time.perf_counter()
        with torch.cuda.device(devs[0]):
            cnt = [int(n_obs)] if cats is None else [int(c) for c in counts]
            means = _engine.csc_column_means(X_csc, groups, n_groups, cnt, np_dtype=mean_dtype)
        given = _means_from_chains(list(means), counts, cats, True)
        need_means = False
        tm["reference_pass"] = _time.perf_counter() - t0

    if need_means and X_fortran is not None and groups is None and not f64_means:
        # F-ordered input, all-cell mean: whole columns are needed -- formed on the first GPU before the shards start
        # (per-category means go through X[rows, :], which numpy returns C-ordered: the chains below)
        t0 = _time.perf_counter()
        with torch.cuda.device(devs[0]):
            given = _engine.fortran_column_means(X_fortran, np_dtype=mean_dtype)[np.newaxis, :]
        need_means = False
        tm["reference_pass"] = _time.perf_counter() - t0

    barrier = threading.Barrier(len(shards)) if multi else None
    chained = [threading.Event() for _ in shards]  # shard k's reference-order accumulators are on the host
    ref_box = {}
    errors = []

    def wait_for(ev):
        while not ev.wait(0.05):
            if errors:
                raise threading.BrokenBarrierError

    def per_row_bytes(plan):
        if sp.issparse(X):
            per_row = max(1.0, X.nnz / max(n_obs, 1)) * (esz + 4) + 8 + 4 * plan.n_windows + 64
        else:
            per_row = n_vars * esz + 4 * plan.n_windows + 64
        per_row += 8 * plan.n_windows  # float64 window scratch of the chromosome-group fallback (rows that exceed LDS)
        if calculate_gene_values:  # float64 gene matrix + float64 windows + covered-gene means
            per_row += 8 * (2 * n_vars + plan.n_windows) + 4 * plan.n_windows
        return per_row

    def run_shard(s: _Shard):
        """Everything one GPU does, on the calling thread (its own thread when there are several shards)."""
        t_sh = _time.perf_counter()
        ent, plan = (ent0, plan0) if s.index == 0 else _checkout_plan(var_chrom, var_start, window_size, step,
                                                                      exclude_chromosomes, s.device)
        n_rows = s.g1 - s.g0
        # pieces of a slab: a few chunks each (~2 GB of input), copied by a helper thread while earlier ones compute
        per_row = per_row_bytes(plan)
        piece_bytes = float(os.environ.get("ICV_PIECE_BYTES", 2e9))  # (developer knob: pipeline granularity)
        piece_rows = max(chunksize, int(piece_bytes // max(per_row, 1)) // chunksize * chunksize)
        # row slabs (multiples of chunksize) sized to fit this shard's share of the GPU's free HBM, next to the packed
        # results of the pieces on their way back (worst case 12 bytes per window, three pieces at a time)
        free_b = _engine.free_hbm_bytes()
        packed = 3 * min(piece_rows, max(n_rows, 1)) * plan.n_windows * 12
        slab_rows = int(max(0.45 * free_b / s.share - packed, 0) // per_row)
        slab_rows = max(chunksize, slab_rows // chunksize * chunksize)
        slabs = [(r, min(n_rows, r + slab_rows)) for r in range(0, max(n_rows, 1), slab_rows)] if n_rows else []
        t_h2d = [0.0]
        streams = {}
        drain = None

        def retire(k):
            ss = streams.pop(k)
            ss.close()
            t_h2d[0] += ss.h2d_seconds
            if getattr(ss, "pack_stats", None):
                for kk, vv in ss.pack_stats.items():
                    s.tm["pack_" + kk] = vv if kk == "threads" else s.tm.get("pack_" + kk, 0.0) + vv

        def slab_stream(i):
            """Start (or return) the upload of slab i; the previous slab is released first (one slab resident)."""
            if i not in streams:
                for k in list(streams):
                    retire(k)
                r0, r1 = slabs[i]
                # (the parent's arrays are read in place: no host copy of the shard or the slab)
                streams[i] = _engine.SlabStream(X, tdtype, piece_rows, s.g0 + r0, s.g0 + r1,
                                                host_pack_threads=max(2, _engine._default_pack_threads() // len(shards)))
                s.tm["sparse_upload"] = bool(streams[i].sparse_upload)
            return streams[i]

        try:
            if f64_means:
                # opt-in: float64 column sums of this shard's rows, all shards at once; the host adds the shards' sums
                t0 = _time.perf_counter()
                sums = torch.zeros((n_groups, n_vars), dtype=torch.float64, device="cuda")
                for i, (s0, _) in enumerate(slabs):
                    ss = slab_stream(i)
                    for r0, r1 in ss.pieces():
                        rg = None if groups is None else groups[s.g0 + s0 + r0: s.g0 + s0 + r1]
                        _engine.column_sums(ss.dm, rg, n_groups, sums, r0, r1)
                    ss = None
                s.accs = sums.cpu().numpy()
                if multi:
                    barrier.wait()
                    if s.index == 0:
                        ref_box["ref"] = _means_from_sums([sh.accs for sh in shards], counts, cats, mean_dtype)
                    barrier.wait()
                    ref = ref_box["ref"]
                else:
                    ref = _means_from_sums([s.accs], counts, cats, mean_dtype)
                s.tm["reference_pass"] = _time.perf_counter() - t0
            elif need_means:
                t0 = _time.perf_counter()
                if slabs:
                    slab_stream(0)  # the upload starts now, whatever this shard has to wait for
                accs = [None] * n_groups
                if s.index > 0:  # continue the chains of the rows before this shard
                    wait_for(chained[s.index - 1])
                    accs = [torch.from_numpy(a).cuda() for a in shards[s.index - 1].accs]
                for i, (s0, _) in enumerate(slabs):
                    ss = slab_stream(i)
                    for r0, r1 in ss.pieces():
                        for gi in range(n_groups):
                            rows, n_g = None, counts
                            if groups is not None:
                                rows = np.nonzero(groups[s.g0 + s0 + r0: s.g0 + s0 + r1] == gi)[0]
                                n_g = counts[gi]
                            accs[gi] = _engine.column_chain(ss.dm, accs[gi], rows, n_g, r0, r1)
                    ss = None
                s.accs = [(a.cpu().numpy() if a is not None else np.zeros(n_vars, dtype=mean_dtype)) for a in accs]
                chained[s.index].set()
                if multi:
                    barrier.wait()  # every shard's accumulators are on the host
                    if s.index == 0:
                        ref_box["ref"] = _means_from_chains(shards[-1].accs, counts, cats, sp.issparse(X))
                    barrier.wait()
                    ref = ref_box["ref"]
                else:
                    ref = _means_from_chains(s.accs, counts, cats, sp.issparse(X))
                s.tm["reference_pass"] = _time.perf_counter() - t0
            else:
                ref = given
            flags = 0
            if ref.shape[0] == 1:
                ref_lo = torch.from_numpy(np.ascontiguousarray(ref[0].astype(compute))).cuda()
                ref_hi = None
            else:
                ref_lo = torch.from_numpy(np.ascontiguousarray(np.min(ref, axis=0).astype(compute))).cuda()
                ref_hi = torch.from_numpy(np.ascontiguousarray(np.max(ref, axis=0).astype(compute))).cuda()
                if x_is_int:
                    flags |= _lib.ICV_FLAG_TRUNC_TO_INT
                elif X.dtype in (np.float32, np.float16) and compute == np.float64:
                    flags |= _lib.ICV_FLAG_ROUND_F32

            t0 = _time.perf_counter()
            drain = _engine.CsrDrain(n_rows, plan.n_windows)  # packs and copies back finished pieces behind the kernels
            for i in range(len(slabs)):
                ss = slab_stream(i)  # a single slab that a reference pass has already brought in is not uploaded again
                for r0, r1 in ss.pieces():
                    # (calculate_gene_values: the smoothing launch also writes its float64 windows; the gene layer of
                    # the piece is formed from them in one kernel -- pieces are whole chunks, so the piece's thresholds
                    # are those of its rows)
                    res = _engine.run_hot_path(plan, ss.dm, ref_lo, ref_hi, lfc_clip=lfc_clip,
                                               dynamic_threshold=dynamic_threshold, chunksize=chunksize, flags=flags,
                                               row0=r0, row1=r1, apply=False, windows=calculate_gene_values)
                    # step 5b + csr_matrix(x_res) on the device (keep-mask, row offsets, fill); only the row offsets
                    # and the packed entries are read back by the drain
                    drain.submit(_engine.threshold_csr(plan, ss.dm, ref_lo, ref_hi, res, lfc_clip=lfc_clip,
                                                       chunksize=chunksize, flags=flags, row0=r0, row1=r1))
                    if calculate_gene_values and r1 > r0:
                        gv = _engine.gene_values_from_windows(plan, res.windows, thr=res.thr, chunksize=chunksize,
                                                              n_vars=n_vars)
                        s.gene_pieces.append(gv.cpu().numpy())
                        del gv
                    del res
                ss = None
            for k in list(streams):
                retire(k)
            s.tm["stream_and_kernels"] = _time.perf_counter() - t0
            tp = _time.perf_counter()
            s.result = drain.finish(arrays=multi)
            s.tm["h2d"] = t_h2d[0]
            s.tm["csr_pack_d2h"] = drain.busy_seconds        # mostly hidden behind the uploads and kernels
            s.tm["csr_pack_d2h_tail"] = _time.perf_counter() - tp  # what was left after the last kernel was launched
            s.tm["rows"] = n_rows
            s.tm["device"] = s.device
            s.tm["kernel"] = plan.last_kernel()  # _lib.ICV_KERNEL_*: which smoothing kernel the last piece took
            s.tm["total"] = _time.perf_counter() - t_sh
        finally:
            # error or not: stop the helper threads, release their device buffers (a failed call used to
            # leave the drain thread blocked on its queue with the GPU tensors pinned)
            for k in list(streams):
                streams.pop(k).close(cancel=True)
            if drain is not None:
                drain.close()
            if plan is not plan0:
                _checkin_plan(ent, plan)

    def shard_thread(s: _Shard):
        try:
            with torch.cuda.device(s.device):
                # a stream of its own: shards that share a GPU overlap, and nothing queues behind the caller's work
                with torch.cuda.stream(torch.cuda.Stream(s.device)):
                    run_shard(s)
                    torch.cuda.current_stream().synchronize()
        except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
            errors.append((s.index, e))
            if barrier is not None:
                barrier.abort()

    # the plans come from the cache the resident path uses (one per concurrent user and device, handed back at the end):
    # planning the gene order costs ~6 ms of host time at 20 000 genes -- 4 % of a 200 000-cell call -- the first time only
    t0 = _time.perf_counter()
    ent0, plan0 = _checkout_plan(var_chrom, var_start, window_size, step, exclude_chromosomes, shards[0].device)
    tm["plan"] = _time.perf_counter() - t0
    try:
        if plan0.n_without_position:
            log.warning(f"Skipped {plan0.n_without_position} genes because they don't have a genomic position annotated. ")
        if multi:
            threads = [threading.Thread(target=shard_thread, args=(s,), daemon=True) for s in shards]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            if errors:
                real = [e for _, e in sorted(errors, key=lambda x: x[0]) if not isinstance(e, threading.BrokenBarrierError)]
                raise (real[0] if real else errors[0][1])
        else:
            with torch.cuda.device(shards[0].device):
                run_shard(shards[0])
        chr_pos = dict(plan0.chr_pos)
        n_windows = plan0.n_windows
    finally:
        _checkin_plan(ent0, plan0)

    # ---- vstack of the shards (:137) --------------------------------------------------------------------------------
    if multi:
        t0 = _time.perf_counter()
        res_mat = _concat_csr([s.result for s in shards], n_obs, n_windows)
        tm["concat"] = _time.perf_counter() - t0
        tm["shards"] = [{k: (round(float(v), 4) if isinstance(v, float) else v) for k, v in s.tm.items()} for s in shards]
        for k in ("reference_pass", "stream_and_kernels", "h2d", "csr_pack_d2h", "csr_pack_d2h_tail"):
            vals = [s.tm[k] for s in shards if k in s.tm]
            if vals:
                tm[k] = max(vals)  # the shards run concurrently: the slowest one is what the call waits for
    else:
        res_mat = shards[0].result
        tm.update({k: v for k, v in shards[0].tm.items() if k not in ("rows", "device", "total")})
    tm["kernel"] = shards[0].tm.get("kernel", 0)

    per_gene_mtx = None
    if calculate_gene_values:
        pieces = [p for s in shards for p in s.gene_pieces]
        per_gene_mtx = np.vstack(pieces) if pieces else np.zeros((0, n_vars))
    tm["total"] = _time.perf_counter() - t_start

    if inplace:
        adata.obsm[f"X_{key_added}"] = res_mat
        adata.uns[key_added] = {"chr_pos": chr_pos}
        if calculate_gene_values:
            adata.layers[f"gene_values_{key_added}"] = per_gene_mtx
    else:
        return chr_pos, res_mat, per_gene_mtx


def _concat_csr(parts, n_rows, n_cols):
    """Row-wise concatenation of the shards' CSR arrays into one matrix; the large copies run on one thread per
    shard (numpy releases the GIL for them)."""
    nnz_off = np.concatenate([[0], np.cumsum([int(ip[-1]) for ip, _, _ in parts])]).astype(np.int64)
    row_off = np.concatenate([[0], np.cumsum([len(ip) - 1 for ip, _, _ in parts])]).astype(np.int64)
    assert row_off[-1] == n_rows, (row_off[-1], n_rows)
    indptr = np.empty(n_rows + 1, dtype=np.int64)
    indices = np.empty(int(nnz_off[-1]), dtype=np.int32)
    data = np.empty(int(nnz_off[-1]), dtype=np.float64)
    indptr[0] = 0

    def copy(k):
        ip, ix, dv = parts[k]
        n = int(ip[-1])
        indptr[row_off[k] + 1: row_off[k + 1] + 1] = ip[1:] + nnz_off[k]
        indices[nnz_off[k]: nnz_off[k] + n] = ix[:n]
        data[nnz_off[k]: nnz_off[k] + n] = dv[:n]

    threads = [threading.Thread(target=copy, args=(k,)) for k in range(len(parts))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return sp.csr_matrix((data, indices, indptr), shape=(n_rows, n_cols))

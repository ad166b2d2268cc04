"""``tl.infercnv`` -- drop-in for ``infercnvpy.tl.infercnv`` on AMD MI355X.

Same keyword-only signature, defaults, error behaviour and AnnData side effects as the reference
driver (icbi-lab/infercnvpy ``src/infercnvpy/tl/_infercnv.py:18-161``).  The numeric work of the
reference's ``_infercnv_chunk`` (:411-457) and ``_get_reference`` (:359-408) runs in hand-written
gfx950 kernels behind the C ABI ``include/infercnv_hip.h``; this module only validates, plans the
gene order, moves data and writes the result fields.
"""
from __future__ import annotations

import logging
from collections.abc import Sequence

import numpy as np
import scipy.sparse as sp

from .. import _engine, _lib
from .._plan import GenePlan

log = logging.getLogger("infercnvpy_amd")


def _as_float_kind(dtype) -> str:
    return np.dtype(dtype).kind


def _reference_rows(X, obs, reference_key, reference_cat, reference, n_vars, dm_pieces):
    """R x G reference profile as a host float array (reference ``_get_reference``, :359-408).

    Means are computed on the GPU (float64 column sums / count) and rounded to the dtype numpy
    would have produced (float32 matrix -> float32 mean, everything else -> float64).
    ``dm_pieces()`` yields ``(device matrix, row0, row1, global_row0)``: row ranges resident in HBM, in order.
    """
    if reference is not None:
        ref = np.asarray(reference)
        if isinstance(ref, np.matrix):
            ref = np.asarray(ref)
    else:
        mean_dtype = np.float32 if X.dtype == np.float32 else np.float64
        if reference_key is None or reference_cat is None:
            log.warning("Using mean of all cells as reference. For better results, provide either "
                        "`reference`, or both `reference_key` and `reference_cat`. ")
            sums, n = None, 0
            for dm, r0, r1, _ in dm_pieces():
                sums = _engine.column_sums(dm, None, 1, sums, r0, r1)
                n += r1 - r0
            ref = (sums / n).cpu().numpy().astype(mean_dtype)
        else:
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
            dup = len(set(cats.tolist())) != len(cats)
            sums = None
            for dm, r0, r1, g0 in dm_pieces():
                sums = _engine.column_sums(dm, groups[g0: g0 + (r1 - r0)], len(cats), sums, r0, r1)
            sums = sums.cpu().numpy()
            if dup:  # same label listed twice: rows repeat
                first = {c: i for i, c in reversed(list(enumerate(cats.tolist())))}
                sums = np.vstack([sums[first[c]] for c in cats.tolist()])
            ref = (sums / counts[:, None]).astype(mean_dtype)
    if ref.ndim == 1:
        ref = ref[np.newaxis, :]
    if ref.shape[1] != n_vars:
        raise ValueError("Reference must match the number of genes in AnnData. ")
    return ref


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
    _timings: dict | None = None,
):
    """Infer copy number variation by averaging gene expression over genomic regions (GPU).

    Parameters and return value as the reference function (``tl/_infercnv.py:18-96``).  ``n_jobs`` is
    accepted for compatibility and ignored (cells are processed by one workgroup each on the GPU;
    ``chunksize`` keeps its numerical meaning: the noise threshold is the standard deviation of
    each ``chunksize``-cell chunk, reference :449-451).  ``_timings`` (not part of the reference API): a dict
    that receives the wall-clock seconds of the stages (plan, host -> HBM copy, kernels, CSR pack + copy back).

    Precision of ``X_cnv``: every window is accumulated, centred and compared with the noise threshold in float64
    (as the reference's ``np.convolve`` is) and stored on the device as float32; the CSR values are those float32
    numbers widened to float64.  For float64 / integer input the reference keeps full float64 values, so entries
    differ from it by up to half a float32 ulp (|x| <= lfc_clip = 3: 1.2e-7 absolute; the tests bound 1e-6); the
    zero pattern is exact (ties with the threshold are re-decided in float64).

    Data movement: the rows are copied to HBM in pieces of a few chunks by a helper thread on a side stream
    while the pieces that have landed are smoothed (reference means: summed); X_cnv is packed to CSR on the
    GPU from the un-thresholded result and a keep-mask (x_res is never rewritten) and only the packed arrays
    cross PCIe on the way back.
    """
    import time as _time

    tm = _timings if _timings is not None else {}
    t_start = _time.perf_counter()
    if not adata.var_names.is_unique:
        raise ValueError("Ensure your var_names are unique!")
    if {"chromosome", "start", "end"} - set(adata.var.columns) != set():
        raise ValueError(
            "Genomic positions not found. There need to be `chromosome`, `start`, and `end` columns in `adata.var`. ")
    _lib.load()  # fail loudly before doing any work if the HIP extension is missing

    plan = GenePlan(adata.var["chromosome"].to_numpy(), adata.var["start"].to_numpy(),
                    window_size=window_size, step=step, exclude_chromosomes=exclude_chromosomes)
    if plan.n_without_position:
        log.warning(f"Skipped {plan.n_without_position} genes because they don't have a genomic position annotated. ")

    X = adata.X if layer is None else adata.layers[layer]
    if isinstance(X, np.matrix):
        X = np.asarray(X)
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
    x_kind = X.dtype.kind
    x_is_int = x_kind in "iub"
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

    # ---- row slabs (multiples of chunksize) sized to fit HBM -----------------------------------
    free_b, _ = torch.cuda.mem_get_info()
    esz = 4 if compute == np.float32 else 8
    if sp.issparse(X):
        per_row = max(1.0, X.nnz / max(n_obs, 1)) * (esz + 4) + 8 + 4 * plan.n_windows + 64
    else:
        per_row = n_vars * esz + 4 * plan.n_windows + 64
    per_row += 8 * plan.n_windows  # float64 window scratch of the chromosome-group fallback (rows that exceed LDS)
    if calculate_gene_values:  # float64 gene matrix + float64 windows + covered-gene means
        per_row += 8 * (2 * n_vars + plan.n_windows) + 4 * plan.n_windows
    slab_rows = int((0.45 * free_b) // per_row)
    slab_rows = max(chunksize, slab_rows // chunksize * chunksize)
    bounds = [(r, min(n_obs, r + slab_rows)) for r in range(0, max(n_obs, 1), slab_rows)] if n_obs else []

    # pieces of a slab: a few chunks each (~2 GB of input), copied by a helper thread while earlier ones compute
    piece_rows = max(chunksize, int(2e9 // max(per_row, 1)) // chunksize * chunksize)
    tm["plan"] = _time.perf_counter() - t_start
    t_h2d = [0.0]
    streams = {}

    def slab_stream(i):
        """Start (or return) the upload of slab i; only one slab is resident at a time."""
        if i not in streams:
            for k in list(streams):
                t_h2d[0] += streams.pop(k).h2d_seconds
            r0, r1 = bounds[i]
            rows = X if (r0 == 0 and r1 == n_obs) else X[r0:r1]  # slicing a CSR matrix copies it
            streams[i] = _engine.SlabStream(rows, tdtype, piece_rows)
        return streams[i]

    def dm_pieces():
        for i, (g0, _) in enumerate(bounds):
            ss = slab_stream(i)
            for r0, r1 in ss.pieces():
                yield ss.dm, r0, r1, g0 + r0

    need_means = reference is None
    t0 = _time.perf_counter()
    ref = _reference_rows(X, adata.obs, reference_key, reference_cat, reference, n_vars, dm_pieces)
    if need_means:
        tm["reference_pass"] = _time.perf_counter() - t0
    n_ref = ref.shape[0]
    flags = 0
    if n_ref == 1:
        ref_lo = torch.from_numpy(np.ascontiguousarray(ref[0].astype(compute))).cuda()
        ref_hi = None
    else:
        ref_lo = torch.from_numpy(np.ascontiguousarray(np.min(ref, axis=0).astype(compute))).cuda()
        ref_hi = torch.from_numpy(np.ascontiguousarray(np.max(ref, axis=0).astype(compute))).cuda()
        if x_is_int:
            flags |= _lib.ICV_FLAG_TRUNC_TO_INT
        elif X.dtype in (np.float32, np.float16) and compute == np.float64:
            flags |= _lib.ICV_FLAG_ROUND_F32

    gene_pieces = []
    t0 = _time.perf_counter()
    drain = _engine.CsrDrain(n_obs, plan.n_windows)  # packs and copies back finished pieces behind the kernels
    for i in range(len(bounds)):
        ss = slab_stream(i)  # a single slab that a reference pass has already brought in is not uploaded again
        thrs = []
        for r0, r1 in ss.pieces():
            res = _engine.run_hot_path(plan, ss.dm, ref_lo, ref_hi, lfc_clip=lfc_clip,
                                       dynamic_threshold=dynamic_threshold, chunksize=chunksize, flags=flags,
                                       row0=r0, row1=r1, apply=False)
            drain.submit(_engine.threshold_mask(plan, ss.dm, ref_lo, ref_hi, res, lfc_clip=lfc_clip,
                                                chunksize=chunksize, flags=flags, row0=r0, row1=r1))
            if res.thr is not None:
                thrs.append(res.thr)
            del res
        if calculate_gene_values:
            thr_all = torch.cat(thrs) if thrs else None
            gv = _engine.gene_values(plan, ss.dm, ref_lo, ref_hi, lfc_clip=lfc_clip, thr=thr_all,
                                     chunksize=chunksize, flags=flags)
            gene_pieces.append(gv.cpu().numpy())
    for k in list(streams):
        t_h2d[0] += streams.pop(k).h2d_seconds
    tm["stream_and_kernels"] = _time.perf_counter() - t0
    tp = _time.perf_counter()
    res_mat = drain.finish()
    tm["h2d"] = t_h2d[0]
    tm["csr_pack_d2h"] = drain.busy_seconds        # mostly hidden behind the uploads and kernels
    tm["csr_pack_d2h_tail"] = _time.perf_counter() - tp  # what was left after the last kernel was launched

    chr_pos = dict(plan.chr_pos)
    per_gene_mtx = None
    if calculate_gene_values:
        per_gene_mtx = np.vstack(gene_pieces) if gene_pieces else np.zeros((0, n_vars))
    plan.close()
    tm["total"] = _time.perf_counter() - t_start

    if inplace:
        adata.obsm[f"X_{key_added}"] = res_mat
        adata.uns[key_added] = {"chr_pos": chr_pos}
        if calculate_gene_values:
            adata.layers[f"gene_values_{key_added}"] = per_gene_mtx
    else:
        return chr_pos, res_mat, per_gene_mtx

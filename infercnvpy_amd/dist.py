"""Multi-GPU execution of the hot path: one process per GPU, cells sharded by rows.

The reference's only parallelism is a process pool over 5000-cell row chunks
(icbi-lab/infercnvpy ``tl/_infercnv.py:120-135``); rows are independent through steps 1-4 of the
chunk kernel, so the shard unit here is the same row chunk.  Two quantities couple cells:

* the reference profile when it is a mean over cells (``_get_reference``, :385/:400): every rank
  accumulates float64 column sums of its rows on its GPU and ONE all-reduce (RCCL over xGMI with
  the ``nccl`` backend, ``gloo`` in the CPU tests) of the ``[R, G]`` sums plus ``[R]`` counts
  gives every rank the same means;
* the noise threshold, a population std over each ``chunksize``-row chunk (:449-451): shards are
  aligned to ``chunksize`` by default, so every chunk lives on one rank and no collective is
  needed; for unaligned shards the per-chunk ``(n, sum, sum of squares)`` triples are all-reduced.

Nothing else is exchanged: outputs stay sharded, the host concatenates (``vstack``, :137).
"""
from __future__ import annotations

import math

import numpy as np


def _dist():
    import torch.distributed as dist

    return dist


def world():
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_obs: int, world_size: int, chunksize: int, align: bool = True):
    """Contiguous row ranges [(r0, r1)] per rank.

    ``align=True``: boundaries are multiples of ``chunksize`` (the chunks of reference :123 are
    dealt out as evenly as possible), so the per-chunk std never crosses ranks.
    """
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if align:
        n_chunks = math.ceil(n_obs / chunksize) if n_obs else 0
        base, extra = divmod(n_chunks, world_size)
        bounds, c0 = [], 0
        for r in range(world_size):
            c1 = c0 + base + (1 if r < extra else 0)
            bounds.append((min(n_obs, c0 * chunksize), min(n_obs, c1 * chunksize)))
            c0 = c1
        return bounds
    base, extra = divmod(n_obs, world_size)
    bounds, r0 = [], 0
    for r in range(world_size):
        r1 = r0 + base + (1 if r < extra else 0)
        bounds.append((r0, r1))
        r0 = r1
    return bounds


def all_reduce_sum_(tensor, group=None):
    """In-place sum over ranks (no-op for a single process)."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def reference_means(local_sums, local_counts, out_dtype, group=None):
    """All-reduce per-group float64 column sums and counts; return the R x G means (numpy).

    ``local_sums``: torch float64 ``[R, G]`` on this rank's device (what ``icv_colsum`` produced for
    this rank's rows); ``local_counts``: length-R integer counts of this rank's rows per group.
    """
    import torch

    counts = torch.as_tensor(np.asarray(local_counts, dtype=np.float64)).to(local_sums.device)
    # one message: sums and counts travel together
    packed = torch.cat([local_sums.reshape(-1), counts])
    all_reduce_sum_(packed, group)
    r = local_sums.shape[0]
    sums = packed[:-r].reshape(local_sums.shape)
    cnt = packed[-r:]
    if bool((cnt == 0).any()):
        raise ValueError("a reference category has no cells on any rank")
    return (sums / cnt[:, None]).cpu().numpy().astype(out_dtype)


def chunk_moments(cell_stats, global_row0: int, chunksize: int, n_chunks_global: int):
    """Per global chunk (rows, sum x, sum x^2) contributed by this rank's rows.

    ``cell_stats``: torch float64 ``[rows, 2]`` (per-cell sum and sum of squares of x_res, as written
    by ``icv_infercnv_smooth``), for the contiguous global rows ``global_row0 ...``.  Deterministic
    (plain reductions over contiguous segments, no atomics).  Returns float64 ``[n_chunks_global, 3]``.
    """
    import torch

    rows = cell_stats.shape[0]
    out = torch.zeros((n_chunks_global, 3), dtype=torch.float64, device=cell_stats.device)
    r = 0
    while r < rows:
        g = global_row0 + r
        k = g // chunksize
        seg = min(rows - r, (k + 1) * chunksize - g)
        if seg == chunksize and (rows - r) >= 2 * chunksize:
            # run of whole chunks: one reshape-reduce
            n_full = (rows - r) // chunksize
            blk = cell_stats[r:r + n_full * chunksize].reshape(n_full, chunksize, 2).sum(dim=1)
            out[k:k + n_full, 0] += chunksize
            out[k:k + n_full, 1:] += blk
            r += n_full * chunksize
            continue
        out[k, 0] += seg
        out[k, 1:] += cell_stats[r:r + seg].sum(dim=0)
        r += seg
    return out


def thresholds_from_moments(moments, n_windows: int, dynamic_threshold: float):
    """thr[k] = dynamic_threshold * population std of chunk k (reference :450) from global moments."""
    import torch

    n = moments[:, 0] * n_windows
    mean = moments[:, 1] / n
    var = torch.clamp((moments[:, 2] - moments[:, 1] * mean) / n, min=0.0)
    return dynamic_threshold * torch.sqrt(var)


def global_thresholds(cell_stats, global_row0, n_obs_global, chunksize, n_windows, dynamic_threshold, group=None):
    """Thresholds of every global chunk, identical on all ranks (one small all-reduce)."""
    n_chunks = max(1, math.ceil(n_obs_global / chunksize))
    m = chunk_moments(cell_stats, global_row0, chunksize, n_chunks)
    all_reduce_sum_(m, group)
    return thresholds_from_moments(m, n_windows, dynamic_threshold)


def run_shard(plan, dm_local, ref_lo, ref_hi=None, *, global_row0=0, n_obs_global=None, lfc_clip=3.0,
              dynamic_threshold=1.5, chunksize=5000, flags=0, group=None):
    """Hot path for this rank's rows of a row-sharded matrix (device resident).

    Chunk-aligned shards (``global_row0 % chunksize == 0``) need no communication here; otherwise
    the smoothing kernel runs first, the chunk moments are all-reduced and the thresholds applied.
    Returns the local :class:`infercnvpy_amd._engine.SmoothResult`.
    """
    import ctypes as C

    from . import _engine, _lib

    rows = dm_local.shape[0]
    n_obs_global = rows if n_obs_global is None else n_obs_global
    aligned = (global_row0 % chunksize == 0) and ((global_row0 + rows) % chunksize == 0
                                                  or global_row0 + rows == n_obs_global)
    if dynamic_threshold is None or aligned:
        return _engine.run_hot_path(plan, dm_local, ref_lo, ref_hi, lfc_clip=lfc_clip,
                                    dynamic_threshold=dynamic_threshold, chunksize=chunksize, flags=flags)
    torch = _engine._torch()
    lib = _lib.load()
    res = _engine.run_hot_path(plan, dm_local, ref_lo, ref_hi, lfc_clip=lfc_clip, dynamic_threshold=None,
                               chunksize=chunksize, flags=flags)
    thr_all = global_thresholds(res.cell_stats, global_row0, n_obs_global, chunksize, plan.n_windows,
                                float(dynamic_threshold), group)
    k0 = global_row0 // chunksize
    k1 = (global_row0 + rows - 1) // chunksize
    thr = thr_all[k0:k1 + 1].contiguous()
    m = dm_local.c_struct()
    _lib.check(lib.icv_apply_threshold(
        plan.handle, C.byref(m), _engine._ptr(ref_lo), _engine._ptr(ref_hi), float(lfc_clip), int(flags),
        _engine._ptr(res.out), res.out.stride(0), _engine._ptr(res.cell_median), _engine._ptr(thr), int(chunksize),
        int(global_row0 % chunksize), _engine._stream_ptr(torch)))
    res.thr = thr
    return res

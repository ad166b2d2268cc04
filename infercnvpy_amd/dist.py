"""Multi-GPU execution of the hot path: one process per GPU, cells sharded by rows.

The reference's only parallelism is a process pool over 5000-cell row chunks
(icbi-lab/infercnvpy ``tl/_infercnv.py:120-135``); rows are independent through steps 1-4 of the
chunk kernel, so the shard unit here is the same row chunk.  Two quantities couple cells:

* the reference profile when it is a mean over cells (``_get_reference``, :385/:400): every rank
  accumulates float64 column sums of its rows on its GPU and ONE all-reduce (RCCL over xGMI with
  the ``nccl`` backend, ``gloo`` in the CPU tests) of the ``[R, G]`` sums plus ``[R]`` counts
  gives every rank the same, correctly rounded means (:func:`reference_means`); the reference's OWN
  bits need its evaluation order, a sequential chain per column: :func:`reference_means_chained`
  passes the accumulators from rank to rank instead (the ranks take turns);
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
    """In-place sum over ranks (no-op for a single process).  Device tensors travel through RCCL with the ``nccl``
    backend; with any other backend (``gloo``: the CPU tests, and several ranks sharing one GPU) through the host."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if tensor.is_cuda and dist.get_backend(group) != "nccl":
            h = tensor.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            tensor.copy_(h)
        else:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def reference_means(local_sums, local_counts, out_dtype, group=None, device_out=False):
    """All-reduce per-group float64 column sums and counts; return the R x G means.

    ``local_sums``: torch float64 ``[R, G]`` on this rank's device (what ``icv_colsum`` produced for
    this rank's rows); ``local_counts``: length-R integer counts of this rank's rows per group.
    Returns a numpy array of ``out_dtype``, or with ``device_out=True`` a torch tensor on the device of
    ``local_sums`` (no host round trip: nothing synchronises with the host; an empty category then gives
    NaN means instead of raising).
    """
    import torch

    counts = torch.as_tensor(np.asarray(local_counts, dtype=np.float64)).to(local_sums.device)
    # one message: sums and counts travel together
    packed = torch.cat([local_sums.reshape(-1), counts])
    all_reduce_sum_(packed, group)
    r = local_sums.shape[0]
    sums = packed[:-r].reshape(local_sums.shape)
    cnt = packed[-r:]
    if device_out:
        tdtype = {"float32": torch.float32, "float64": torch.float64}[np.dtype(out_dtype).name]
        return (sums / cnt[:, None]).to(tdtype)
    if bool((cnt == 0).any()):
        raise ValueError("a reference category has no cells on any rank")
    return (sums / cnt[:, None]).cpu().numpy().astype(out_dtype)


def _p2p(tensor, peer, send, group=None):
    """Blocking send / receive of a device tensor; through the host unless the backend is nccl (RCCL)."""
    dist = _dist()
    if tensor.is_cuda and dist.get_backend(group) != "nccl":
        h = tensor.cpu()
        (dist.send if send else dist.recv)(h, peer, group=group)
        if not send:
            tensor.copy_(h)
    else:
        (dist.send if send else dist.recv)(tensor, peer, group=group)


def _group_ranks(group=None):
    """(rank in the group, group size, global rank of every group member) -- peers of send / recv / broadcast are GLOBAL
    ranks in torch.distributed, whatever the group (the default group's ranks were used for a subgroup)."""
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1, [0]
    size = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if group is None:
        return rank, size, list(range(size))
    return rank, size, [dist.get_global_rank(group, r) for r in range(size)]


def chain_column_groups(n_cols: int, esz: int, n_groups: int):
    """Column ranges [(c0, c1)] of the pipelined chain: ``n_groups`` runs of whole 128-byte lines (the tile unit of
    ``k_colchain``), as even as the lines allow; fewer groups when there are fewer lines."""
    per_line = 128 // esz
    n_lines = -(-n_cols // per_line)
    n_groups = max(1, min(int(n_groups), n_lines))
    out = []
    for t in range(n_groups):
        l0, l1 = t * n_lines // n_groups, (t + 1) * n_lines // n_groups
        out.append((min(n_cols, l0 * per_line), min(n_cols, l1 * per_line)))
    return out


def reference_means_chained(dm_local, counts_global, local_rows=None, group=None, n_col_groups=2):
    """The reference profile in the REFERENCE'S OWN evaluation order over row-sharded ranks (bit-equal to
    ``np.mean(X, axis=0)`` / scipy's CSR mean of the whole matrix, reference :385, :400): a float32 column sum is a
    sequential chain, so rank k continues the accumulators of rank k - 1 (``icv_colchain``) and hands them to rank
    k + 1 -- R x G values point to point -- and the last rank's means are broadcast.

    The chains of different columns are independent, so the hand-over is PIPELINED over ``n_col_groups`` column groups
    (dense matrices): rank k starts group g as soon as rank k - 1 has handed that group over, while rank k - 1 works on
    group g + 1.  With T groups and R ranks the pass costs (T + R - 1) group passes instead of R whole passes (T = 1: the
    ranks simply take turns).  T = 2 is the default: a group of half the columns still fills every CU with a tile, so its
    pass takes half the time of a whole pass; narrower groups do not get faster (one 128-byte line per workgroup is the
    narrowest tile: 0.84 / 0.69 / 0.69 ms for T = 2 / 4 / 8 on 125 000 x 20 000 float32, profiles/r05_chain_column_groups.txt).
    CSR shards are handed over whole (the column tiles of ``k_colchain_csrq`` share one bounds pass).  The float64 :func:`reference_means` is the concurrent alternative:
    correctly rounded, one all-reduce, but not the reference's bits.

    ``dm_local``: this rank's rows (``_engine.DeviceMatrix``); ``counts_global``: rows per category over ALL ranks;
    ``local_rows``: per category the ascending local row indices (None: one category, all rows).  Returns the
    ``[R, G]`` means as a device tensor of the matrix dtype, identical on every rank."""
    import torch

    from . import _engine, _lib

    dist = _dist()
    rank, size, peers = _group_ranks(group)
    n_groups = len(counts_global)
    n_cols = dm_local.shape[1]
    is_csr = dm_local.format == _lib.ICV_CSR
    esz = 4 if dm_local.dtype == torch.float32 else 8
    col_groups = [(0, n_cols)] if (is_csr or size == 1) else chain_column_groups(n_cols, esz, n_col_groups)
    device = getattr(dm_local, "device", "cuda")
    accs = torch.zeros((n_groups, n_cols), dtype=dm_local.dtype, device=device)
    for c0, c1 in col_groups:
        whole = c0 == 0 and c1 == n_cols
        if size > 1 and rank > 0:
            buf = accs if whole else torch.empty((n_groups, c1 - c0), dtype=accs.dtype, device=device)
            _p2p(buf, peers[rank - 1], send=False, group=group)
            if not whole:
                accs[:, c0:c1] = buf
        for g in range(n_groups):
            rows = None if local_rows is None else local_rows[g]
            if dm_local.shape[0] and (rows is None or len(rows)):
                _engine.column_chain(dm_local, accs[g], rows, int(counts_global[g]), cols=None if whole else (c0, c1))
        if size > 1 and rank < size - 1:
            _p2p(accs if whole else accs[:, c0:c1].contiguous(), peers[rank + 1], send=True, group=group)
    means = torch.stack([_engine.chain_mean(accs[g], int(counts_global[g]), is_csr) for g in range(n_groups)])
    if size > 1:
        if means.is_cuda and dist.get_backend(group) != "nccl":
            h = means.cpu()
            dist.broadcast(h, peers[size - 1], group=group)
            means.copy_(h)
        else:
            dist.broadcast(means, peers[size - 1], group=group)
    return means


def reference_means_blocks(dm_local, n_total, group=None, n_col_groups=4, stats=None):
    """The all-cell reference profile of a row-sharded DENSE float32 matrix in the reference's own evaluation order
    (bit-equal to ``np.mean(X, axis=0)`` of the whole matrix, reference :385, and to :func:`reference_means_chained`)
    WITHOUT the ranks taking turns on their rows (``csrc/icv_kernel_blocks.hpp``; proofs: ``tests/exact_chain_proto.py``).

    Inside one binade a float32 chain is integer arithmetic, so a block of rows whose chain stays in its binade is one
    integer ``Q`` that can be formed at any time -- it only needs the binade of its start, from a float64 estimate:

    1. every rank adds the float64 column totals of its rows                       (one pass, concurrent)
    2. ONE all-gather of ``[G]`` float64 per rank: rank k's start estimate = the totals of ranks < k
    3. every rank forms its block records                                          (one pass, concurrent)
    4. the exact float32 chain values travel rank to rank; a rank only SCANS its records (one word per 1024 rows and
       column) and replays the few blocks that cannot be summarised (0.04-0.5 % on ranks > 0) -- pipelined over
       ``n_col_groups`` column groups like the chained form -- and the last rank broadcasts the means.

    Two passes over the shard instead of one, but both concurrent on all ranks: at 8 ranks the means cost ~2 passes of one
    rank's rows + a scan, where the chained form costs (T + R - 1) / T = 4.5 passes.  Other inputs (CSR,
    float64, categories) keep :func:`reference_means_chained`."""
    import torch

    from . import _engine, _lib

    dist = _dist()
    rank, size, peers = _group_ranks(group)
    if dm_local.format != _lib.ICV_DENSE or dm_local.dtype != torch.float32:
        raise ValueError("reference_means_blocks: dense float32 shards (use reference_means_chained)")
    n_cols = dm_local.shape[1]
    device = getattr(dm_local, "device", "cuda")
    have_rows = dm_local.shape[0] > 0
    cb = _engine.ChainBlocks(dm_local) if have_rows else None
    total = cb.sums() if have_rows else torch.zeros(n_cols, dtype=torch.float64, device=device)
    if size > 1:
        on_host = total.is_cuda and dist.get_backend(group) != "nccl"
        t = total.cpu() if on_host else total
        gathered = [torch.empty_like(t) for _ in range(size)]
        dist.all_gather(gathered, t, group=group)
        est = torch.zeros_like(t)
        for k in range(rank):  # (in rank order: every rank derives the same estimates)
            est = est + gathered[k]
        est = est.to(device) if on_host else est
    else:
        est = None
    if have_rows:
        cb.records(est)
    acc = torch.zeros(n_cols, dtype=torch.float32, device=device)
    col_groups = [(0, n_cols)] if size == 1 else chain_column_groups(n_cols, 4, n_col_groups)
    for c0, c1 in col_groups:
        if size > 1 and rank > 0:
            buf = torch.empty(c1 - c0, dtype=torch.float32, device=device)
            _p2p(buf, peers[rank - 1], send=False, group=group)
            acc[c0:c1] = buf
        if have_rows:
            cb.scan(acc, cols=(c0, c1))
        if size > 1 and rank < size - 1:
            _p2p(acc[c0:c1].contiguous(), peers[rank + 1], send=True, group=group)
    means = _engine.chain_mean(acc, int(n_total), False)[None, :]
    if size > 1:
        if means.is_cuda and dist.get_backend(group) != "nccl":
            h = means.cpu()
            dist.broadcast(h, peers[size - 1], group=group)
            means = h.to(device)
        else:
            dist.broadcast(means, peers[size - 1], group=group)
    if stats is not None and have_rows:
        stats["blocks"] = cb.n_blocks()
        stats["replayed"] = int(cb.replayed.item())
    return means


def reference_means_exact(dm_local, n_total, group=None, stats=None):
    """The all-cell reference profile in the reference's own evaluation order (numpy's / scipy's bits) over a row-sharded
    matrix, by whichever exact form is faster at this group size: the chain by integer blocks
    (:func:`reference_means_blocks`: two concurrent passes + a scan that travels) from 4 ranks on for dense float32 shards,
    the chained accumulators (:func:`reference_means_chained`: one pass, the ranks take turns, pipelined over two column
    groups) below that and for every other input.  With R ranks and T = 2 column groups the chained form costs
    (T + R - 1) / T passes of one rank's rows -- 1.5 at R = 2, 2.5 at R = 4, 4.5 at R = 8 -- against ~2.2 for the blocks
    (DESIGN.md 6).  Same bits either way; returns ``[1, G]`` means of the matrix's dtype on every rank."""
    import torch

    from . import _lib

    _, size, _ = _group_ranks(group)
    if size >= 4 and dm_local.format == _lib.ICV_DENSE and dm_local.dtype == torch.float32:
        return reference_means_blocks(dm_local, n_total, group=group, stats=stats)
    return reference_means_chained(dm_local, [n_total], group=group)


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


def shards_aligned(bounds, n_obs_global: int, chunksize: int) -> bool:
    """True if no ``chunksize``-row chunk of the global matrix is split between two shards."""
    return all(r0 == r1 or (r0 % chunksize == 0 and (r1 % chunksize == 0 or r1 == n_obs_global)) for r0, r1 in bounds)


def agree_aligned(local_aligned: bool, device="cpu", group=None) -> bool:
    """Every rank learns whether ALL ranks hold chunk-aligned shards (one-element all-reduce)."""
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(local_aligned)
    import torch

    flag = torch.tensor([0.0 if local_aligned else 1.0], dtype=torch.float64, device=device)
    all_reduce_sum_(flag, group)
    return float(flag.item()) == 0.0


def run_shard(plan, dm_local, ref_lo, ref_hi=None, *, global_row0=0, n_obs_global=None, lfc_clip=3.0,
              dynamic_threshold=1.5, chunksize=5000, flags=0, group=None, all_bounds=None, out=None, pack=False):
    """Hot path for this rank's rows of a row-sharded matrix (device resident).

    Whether the noise threshold needs communication is a property of the WHOLE partition, and every rank must
    take the same branch (the unaligned path is a collective): with ``all_bounds`` (the row ranges of all
    ranks, e.g. from :func:`shard_bounds`) the decision is computed locally and identically everywhere;
    without it the ranks agree through a one-element all-reduce.  Chunk-aligned partitions need no further
    communication; otherwise the smoothing kernel runs first, the chunk moments of ALL ranks (aligned ones
    included) are all-reduced and the thresholds applied.
    Returns the local :class:`infercnvpy_amd._engine.SmoothResult`; with ``pack=True`` ``(result, PackedCsr)``: the
    thresholds are applied while ``X_cnv`` is packed to CSR on the device (``_engine.threshold_csr``; ``result.out``
    stays un-thresholded), as the public call does.
    """
    import ctypes as C

    from . import _engine, _lib

    rows = dm_local.shape[0]
    n_obs_global = rows if n_obs_global is None else n_obs_global
    if all_bounds is not None:
        aligned = shards_aligned(all_bounds, n_obs_global, chunksize)
    else:
        aligned = shards_aligned([(global_row0, global_row0 + rows)], n_obs_global, chunksize)
        if dynamic_threshold is not None:
            aligned = agree_aligned(aligned, ref_lo.device if hasattr(ref_lo, "device") else "cpu", group)
    if dynamic_threshold is None or aligned:
        res = _engine.run_hot_path(plan, dm_local, ref_lo, ref_hi, lfc_clip=lfc_clip,
                                   dynamic_threshold=dynamic_threshold, chunksize=chunksize, flags=flags, out=out,
                                   apply=not pack)
        if not pack:
            return res
        return res, _engine.threshold_csr(plan, dm_local, ref_lo, ref_hi, res, lfc_clip=lfc_clip, chunksize=chunksize,
                                          flags=flags)
    torch = _engine._torch()
    lib = _lib.load()
    res = _engine.run_hot_path(plan, dm_local, ref_lo, ref_hi, lfc_clip=lfc_clip, dynamic_threshold=None,
                               chunksize=chunksize, flags=flags, cell_stats=True, out=out)
    thr_all = global_thresholds(res.cell_stats, global_row0, n_obs_global, chunksize, plan.n_windows,
                                float(dynamic_threshold), group)
    if rows == 0:
        res.thr = thr_all[:0]
        return (res, _engine.threshold_csr(plan, dm_local, ref_lo, ref_hi, res, lfc_clip=lfc_clip, chunksize=chunksize,
                                           flags=flags)) if pack else res
    k0 = global_row0 // chunksize
    k1 = (global_row0 + rows - 1) // chunksize
    thr = thr_all[k0:k1 + 1].contiguous()
    res.thr = thr
    if pack:  # the global thresholds decide while the rows are packed
        return res, _engine.threshold_csr(plan, dm_local, ref_lo, ref_hi, res, lfc_clip=lfc_clip, chunksize=chunksize,
                                          row_phase=int(global_row0 % chunksize), flags=flags)
    m = dm_local.c_struct()
    _lib.check(lib.icv_apply_threshold(
        plan.handle, C.byref(m), _engine._ptr(ref_lo), _engine._ptr(ref_hi), float(lfc_clip), int(flags),
        _engine._ptr(res.out), res.out.stride(0), _engine._ptr(res.cell_median), _engine._ptr(thr), int(chunksize),
        int(global_row0 % chunksize), _engine._stream_ptr(torch)))
    return res


# --------------------------------------------------------------------------------------------------
# BASELINE config 5: cell x cell distances and Ward rounds on a matrix sharded by rows
# --------------------------------------------------------------------------------------------------
SUPER = 1024  # rows per super-row = ICV_SUPER_ROWS (the distance kernel's super-tile)
SUPER_SHIFT = 10


class WardLayout:
    """Ownership of the n x n distance matrix: super-rows of 1024 rows dealt out in a folded cyclic order
    (rank 0 .. R-1, R-1 .. 0, 0 .. R-1, ...).  Row g of the upper triangle holds ``n_super - g`` super-tiles, so
    pairing a long row with a short one gives every rank the same number of tiles to compute (to within one row
    of super-tiles) and the same number of matrix rows to hold."""

    def __init__(self, n: int, world_size: int, rank: int, super_rows: int = SUPER, spare: bool = True):
        # super_rows: 1024 for the HIP kernels (ICV_SUPER_ROWS); the CPU tests of the exchanges use small ones
        assert super_rows >= 8 and super_rows & (super_rows - 1) == 0
        self.n, self.world, self.rank = int(n), int(world_size), int(rank)
        self.S = int(super_rows)
        self.shift = self.S.bit_length() - 1
        self.n_super = (self.n + self.S - 1) // self.S
        g = np.arange(self.n_super)
        m = g % (2 * self.world)
        self.owner = np.where(m < self.world, m, 2 * self.world - 1 - m).astype(np.int64)
        # local super-row index of every global super-row on its owner, and per rank the list of its super-rows
        self.local_index = np.zeros(self.n_super, dtype=np.int64)
        self.supers_of = []
        for r in range(self.world):
            mine = np.flatnonzero(self.owner == r)
            self.local_index[mine] = np.arange(len(mine))
            self.supers_of.append(mine)
        self.k = [len(v) for v in self.supers_of]  # super-rows per rank
        self.sr_local = np.where(self.owner == self.rank, self.local_index, -1).astype(np.int32)
        # row stride: a multiple of 4 floats (vector loads) and, when `spare`, n / 2 spare columns (the Ward rounds
        # then write their column updates as dense strips, include/infercnv_hip.h).  Every rank must use the same
        # stride: the column layout is part of the replicated bookkeeping.
        self.ld = ((self.n + (self.n + 1) // 2 if spare else self.n) + 3) // 4 * 4
        self.rows_padded = self.k[self.rank] * self.S
        # first row of every rank's block in the mirror buffer (all ranks' local rows, concatenated)
        self.dest_row0 = np.concatenate([[0], np.cumsum([k * self.S for k in self.k])]).astype(np.int64)

    def lrow(self, rows):
        """Local row index of global rows (on their owners)."""
        rows = np.asarray(rows, dtype=np.int64)
        return self.local_index[rows >> self.shift] * self.S + (rows & (self.S - 1))

    def row_owner(self, rows):
        return self.owner[np.asarray(rows, dtype=np.int64) >> self.shift]

    def columns_of(self, r):
        """Global column index of every (padded) local row of rank r, clipped to n - 1."""
        g = self.supers_of[r]
        cols = (g[:, None] * self.S + np.arange(self.S)[None, :]).reshape(-1)
        return np.minimum(cols, self.n - 1)

    def tile_plan(self):
        """This rank's super-tiles (its super-rows of the upper triangle): row0, col0, offset of the direct
        block in the local matrix, offset of the transposed block in the mirror buffer (row stride
        ``rows_padded``: the columns of the mirror block are this rank's local rows)."""
        row0, col0, dir_off, mir_off = [], [], [], []
        for ly, gy in enumerate(self.supers_of[self.rank]):
            for gx in range(gy, self.n_super):
                d = self.owner[gx]
                row0.append(gy * self.S)
                col0.append(gx * self.S)
                dir_off.append(ly * self.S * self.ld + gx * self.S)
                mir_off.append((self.dest_row0[d] + self.local_index[gx] * self.S) * self.rows_padded + ly * self.S)
        return (np.asarray(row0, dtype=np.int32), np.asarray(col0, dtype=np.int32),
                np.asarray(dir_off, dtype=np.int64), np.asarray(mir_off, dtype=np.int64))


def _group_rank(group, r):
    dist = _dist()
    return dist.get_global_rank(group, r) if group is not None else r


def _host_staged(t, group):
    """Collectives on device tensors go through the host when the backend cannot take them (gloo in the tests)."""
    dist = _dist()
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _all_to_all_rows(send, send_rows, recv_rows, group=None):
    """all_to_all of row blocks of a 2-D tensor: ``send_rows[d]`` consecutive rows go to rank d."""
    import torch

    dist = _dist()
    recv = torch.empty((int(sum(recv_rows)), send.shape[1]), dtype=send.dtype, device=send.device)
    if _host_staged(send, group):
        r_h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r_h, send.cpu().contiguous(), [int(v) for v in recv_rows], [int(v) for v in send_rows],
                               group=group)
        recv.copy_(r_h)
    else:
        dist.all_to_all_single(recv, send.contiguous(), [int(v) for v in recv_rows], [int(v) for v in send_rows],
                               group=group)
    return recv


def _all_reduce_sum(t, group=None):
    dist = _dist()
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def gather_rows(x_local, group=None):
    """All-gather a row-sharded matrix (shards may differ in length) -> (full matrix, row bounds per rank)."""
    import torch

    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x_local, [(0, x_local.shape[0])]
    ws = dist.get_world_size(group)
    staged = _host_staged(x_local, group)
    dev = "cpu" if staged else x_local.device
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(counts, torch.tensor([x_local.shape[0]], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    full = torch.empty((sum(counts), x_local.shape[1]), dtype=x_local.dtype, device=dev)
    bounds, r0 = [], 0
    for c in counts:
        bounds.append((r0, r0 + c))
        r0 += c
    xl = x_local.cpu() if staged else x_local
    if len(set(counts)) == 1:
        dist.all_gather_into_tensor(full, xl.contiguous(), group=group)
    else:  # ragged shards: one broadcast per owner
        for r, (a, b) in enumerate(bounds):
            if r == dist.get_rank(group):
                full[a:b] = xl
            dist.broadcast(full[a:b], src=_group_rank(group, r), group=group)
    return (full.to(x_local.device) if staged else full), bounds


class HipWardSteps:
    """The step kernels of the sharded linkage behind the C ABI (``include/infercnv_hip.h``)."""

    def __init__(self, n, layout):
        import ctypes as C

        from . import _engine, _lib

        self._C, self._engine, self._lib = C, _engine, _lib
        self.lib = _lib.load()
        self.torch = _engine._torch()
        self.layout = layout
        h = C.c_void_p()
        sr = np.ascontiguousarray(layout.sr_local, dtype=np.int32)
        assert layout.S == SUPER, "the HIP kernels own rows in super-rows of ICV_SUPER_ROWS"
        spare = layout.ld % 4 == 0 and layout.ld >= n + (n + 1) // 2  # WardLayout(spare=True)
        _lib.check(self.lib.icv_ward_create(int(n), sr.ctypes.data, int(layout.n_super), SUPER_SHIFT, int(layout.ld),
                                            1 if spare else 0, C.byref(h), self._st()))
        self.handle = h

    def _st(self):
        return self._engine._stream_ptr(self.torch)

    def close(self):
        if self.handle:
            self.lib.icv_ward_destroy(self.handle)
            self.handle = None

    def distances(self, x_all, d_local, mirror):
        row0, col0, dir_off, mir_off = self.layout.tile_plan()
        if not len(row0):
            return
        p = self._engine._ptr
        self._lib.check(self.lib.icv_pairwise_sqeuclidean_tiles(
            p(x_all), x_all.shape[0], x_all.shape[1], x_all.stride(0), len(row0), row0.ctypes.data, col0.ctypes.data,
            dir_off.ctypes.data, mir_off.ctypes.data, p(d_local), d_local.stride(0), p(mirror), mirror.stride(0),
            self._st()))

    def merge(self, d_local, stage, pslot):
        if not self.layout.rows_padded:
            return
        p = self._engine._ptr
        ps = np.ascontiguousarray(pslot, dtype=np.int32)
        self._lib.check(self.lib.icv_ward_merge(self.handle, p(d_local), d_local.stride(0),
                                                p(stage) if stage is not None and stage.numel() else None,
                                                stage.stride(0) if stage is not None and stage.numel() else 0,
                                                ps.ctypes.data, 0, self._st()))
        self.torch.cuda.current_stream().synchronize()  # pslot is a host array

    def gather(self, d_local, rows_t, slots_t, out):
        """out[q][k] = entry of local row rows_t[q] for the cluster of slot slots_t[k] (device index tensors)."""
        p = self._engine._ptr
        self._lib.check(self.lib.icv_ward_gather(self.handle, p(d_local), d_local.stride(0), p(rows_t), rows_t.numel(),
                                                 p(slots_t), slots_t.numel(), p(out), out.stride(0), self._st()))

    def scatter(self, d_local, v, vrow_p):
        p = self._engine._ptr
        vp = np.ascontiguousarray(vrow_p, dtype=np.int32)
        self._lib.check(self.lib.icv_ward_scatter(self.handle, p(d_local), d_local.stride(0), p(v), v.stride(0),
                                                  vp.ctypes.data, len(vp), self._st()))

    def scan(self, d_local):
        if not self.layout.rows_padded:
            return
        self._lib.check(self.lib.icv_ward_scan(self.handle, self._engine._ptr(d_local), d_local.stride(0), self._st()))

    def pack(self, k, device):
        t = self.torch
        nn = t.empty(max(k, 1), dtype=t.int32, device=device)
        dm = t.empty(max(k, 1), dtype=t.float32, device=device)
        self._lib.check(self.lib.icv_ward_pack_nn(self.handle, self._engine._ptr(nn), self._engine._ptr(dm), self._st()))
        return nn, dm

    def unpack(self, nn, dm):
        self._lib.check(self.lib.icv_ward_unpack_nn(self.handle, self._engine._ptr(nn), self._engine._ptr(dm), self._st()))

    def pairs(self, d_local, all_active):
        c = (self._C.c_int32 * 4)()
        self._lib.check(self.lib.icv_ward_pairs(self.handle, self._engine._ptr(d_local) if d_local.numel() else None,
                                                d_local.stride(0), int(bool(all_active)), c, self._st()))
        return int(c[0]), int(c[1]), int(c[2]), int(c[3])

    def round_pairs(self, n_pairs):
        i = np.empty(max(n_pairs, 1), dtype=np.int32)
        j = np.empty(max(n_pairs, 1), dtype=np.int32)
        self._lib.check(self.lib.icv_ward_round_pairs(self.handle, i.ctypes.data, j.ctypes.data))
        return i[:n_pairs].astype(np.int64), j[:n_pairs].astype(np.int64)

    def finish(self, n):
        Z = np.empty((n - 1, 4), dtype=np.float64)
        rounds = self._C.c_int32(0)
        self._lib.check(self.lib.icv_ward_finish(self.handle, Z.ctypes.data, self._C.byref(rounds)))
        return Z, int(rounds.value)


def _unpack_mirror(layout, d_local, recv_from):
    """Blocks below the diagonal of this rank's rows, from the mirror blocks the other ranks (and this one)
    computed as transposes of their tiles above the diagonal.  ``recv_from[s]``: [my padded rows, s's padded rows]."""
    n, S = layout.n, layout.S
    T = S // 8  # the distance kernel's tile
    for lx, gx in enumerate(layout.supers_of[layout.rank]):
        r0 = lx * S
        for s in range(layout.world):
            blk = recv_from[s]
            for ly, gy in enumerate(layout.supers_of[s]):
                if gy > gx:
                    break
                c0, w = gy * S, min(S, n - gy * S)
                if gy < gx:
                    d_local[r0:r0 + S, c0:c0 + w] = blk[r0:r0 + S, ly * S:ly * S + w]
                else:  # the diagonal super-tile: tiles strictly below the diagonal
                    for a in range(1, 8):
                        wa = min(a * T, w)
                        d_local[r0 + a * T:r0 + (a + 1) * T, c0:c0 + wa] = \
                            blk[r0 + a * T:r0 + (a + 1) * T, ly * S:ly * S + wa]


def ward_linkage_sharded(x_local, *, group=None, steps=None, return_rounds=False, super_rows=SUPER):
    """Ward linkage of all cells of a row-sharded ``X_cnv`` (float32 device matrix: this rank's rows).

    The n x n distance matrix is never gathered: every rank owns ``n / R`` of its rows (super-rows of 1024 in a
    folded cyclic order, :class:`WardLayout`) and the Ward rounds run on the shards.

    1. the cells are all-gathered (n x d float32: 4 GB for 200 000 x 5 000);
    2. every rank computes the super-tiles ON OR ABOVE the diagonal of its super-rows on fp32 MFMA tiles
       (``icv_pairwise_sqeuclidean_tiles``: n (n + 1024) d flop in total over all ranks, the same as one GPU) --
       directly into its rows, and transposed into a mirror buffer that one ``all_to_all`` (RCCL over xGMI)
       delivers to the owners of those rows: the off-diagonal blocks travel once;
    3. Ward rounds (``icv_ward_*``): the owner of a merged row receives the partner row (``all_to_all`` of rows),
       computes the new row and its nearest neighbour, sends every rank the new rows' entries for the clusters of
       that rank's rows (``icv_ward_gather`` + ``all_to_all``) for the update of the rows that did not merge; rows whose cached nearest neighbour merged are searched
       again; one small all-reduce per round makes the (neighbour, distance) results of the round known to all
       ranks, which then find the reciprocal pairs redundantly (replicated O(n) bookkeeping, deterministic).

    Arithmetic per entry is that of the one-GPU path, so the result is bit-identical to ``tl.ward_linkage``.
    ``steps``: factory ``(n, layout) -> object`` with the methods of :class:`HipWardSteps`, and ``super_rows``: the
    gloo tests inject a numpy implementation with small super-rows to exercise the exchanges on CPU.  Returns the float64 linkage matrix on every rank.
    """
    import torch

    dist = _dist()
    x_all, bounds = gather_rows(x_local, group)
    n = x_all.shape[0]
    if n < 2:
        raise ValueError("at least two cells are needed for a linkage")
    if len(bounds) == 1:
        from . import _engine

        d2 = _engine.pairwise_sqeuclidean(x_all, spare=True)
        Z, rounds = _engine.ward_linkage(d2, spare=_engine.has_spare_columns(d2))
        return (Z, rounds) if return_rounds else Z
    rank, ws = dist.get_rank(group), dist.get_world_size(group)
    # spare columns for the Ward rounds only if the distance phase (local rows + mirror buffer + received blocks)
    # still fits on EVERY rank: one all-reduce(MIN) of the local verdict
    L = WardLayout(n, ws, rank, super_rows, spare=True)
    fits = 1
    if x_all.is_cuda:
        free_b, _ = torch.cuda.mem_get_info()
        need = 4 * (L.rows_padded * L.ld + 2 * int(L.dest_row0[-1]) * L.rows_padded) + (2 << 30)
        fits = 1 if need < free_b else 0
    flag = torch.tensor([fits], dtype=torch.int32, device=x_all.device)
    if _host_staged(flag, group):
        h = flag.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
        flag.copy_(h)
    else:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        L = WardLayout(n, ws, rank, super_rows, spare=False)
    ops = (steps or HipWardSteps)(n, L)
    dev = x_all.device
    try:
        # ---- distances: own rows directly, mirror blocks to their owners -------------------------------
        d_local = torch.empty((L.rows_padded, L.ld), dtype=torch.float32, device=dev)
        mirror = torch.empty((int(L.dest_row0[-1]), L.rows_padded), dtype=torch.float32, device=dev)
        ops.distances(x_all, d_local, mirror)
        del x_all
        # what rank s sends me is [my padded rows, its padded rows]: a different width per source, so the
        # exchange is one all_to_all per distinct width (flattened to 1-D)
        send_counts = [L.k[d] * L.S * L.rows_padded for d in range(ws)]
        recv_counts = [L.rows_padded * L.k[s] * L.S for s in range(ws)]
        flat = _all_to_all_rows(mirror.reshape(-1, 1), send_counts, recv_counts, group)
        del mirror
        recv_from, o = [], 0
        for s in range(ws):
            recv_from.append(flat[o:o + recv_counts[s]].view(L.rows_padded, L.k[s] * L.S))
            o += recv_counts[s]
        _unpack_mirror(L, d_local, recv_from)
        del flat, recv_from

        # ---- Ward rounds -------------------------------------------------------------------------------
        n_live, n_merges, n_pairs, n_act = n, 0, 0, n
        pi = pj = np.zeros(0, dtype=np.int64)
        cols_of = [torch.from_numpy(L.columns_of(r).astype(np.int32)).to(dev) for r in range(ws)]  # slots
        retry = False
        while n_live > 1:
            if n_pairs > 0:
                oi, oj = L.row_owner(pi), L.row_owner(pj)
                # partner rows j -> the owners of the rows i that absorbed them
                send_idx = [L.lrow(pj[(oj == rank) & (oi == d)]) if d != rank else np.zeros(0, dtype=np.int64)
                            for d in range(ws)]
                recv_n = [int(((oi == rank) & (oj == s)).sum()) if s != rank else 0 for s in range(ws)]
                idx = np.concatenate(send_idx)
                sendbuf = d_local[torch.from_numpy(idx).to(dev)] if len(idx) else d_local[:0]
                stage = _all_to_all_rows(sendbuf, [len(v) for v in send_idx], recv_n, group)
                pslot = np.full(n_pairs, -1, dtype=np.int32)
                base = 0
                for s in range(ws):
                    sel = np.flatnonzero((oi == rank) & (oj == s)) if s != rank else np.zeros(0, dtype=np.int64)
                    pslot[sel] = base + np.arange(len(sel))
                    base += len(sel)
                ops.merge(d_local, stage, pslot)
                del stage, sendbuf
                # the new rows: every rank gets the entries for the clusters of ITS rows (the sender's own share
                # included), gathered through the column layout of the rounds
                mine = pi[oi == rank]
                rows_t = torch.from_numpy(L.lrow(mine)).to(dev)
                parts = []
                for d in range(ws):
                    out_d = torch.empty((len(mine), L.k[d] * L.S), dtype=torch.float32, device=dev)
                    if len(mine) and L.k[d]:
                        ops.gather(d_local, rows_t, cols_of[d], out_d)
                    parts.append(out_d.reshape(-1, 1))
                send_counts = [len(mine) * L.k[d] * L.S for d in range(ws)]
                recv_counts = [int((oi == s).sum()) * L.rows_padded for s in range(ws)]
                v = _all_to_all_rows(torch.cat(parts), send_counts, recv_counts, group)
                vrow_p = np.concatenate([np.flatnonzero(oi == s) for s in range(ws)])  # merge index of every row of v
                if L.rows_padded:
                    ops.scatter(d_local, v.view(n_pairs, L.rows_padded), vrow_p)
                del v, parts
            ops.scan(d_local)
            nn_t, dm_t = ops.pack(n_pairs + n_act, dev)
            _all_reduce_sum(nn_t, group)
            _all_reduce_sum(dm_t, group)
            ops.unpack(nn_t, dm_t)
            n_live, n_merges, n_pairs, n_act = ops.pairs(d_local, False)
            if n_pairs < 1:
                # cached neighbours of tied distances can point in a cycle: list every live row and search again once
                if retry:
                    raise ValueError("ward_linkage: distances are not finite")
                retry = True
                n_live, n_merges, n_pairs, n_act = ops.pairs(d_local, True)
                pi = pj = np.zeros(0, dtype=np.int64)
                if n_pairs < 1:
                    continue
            retry = False
            pi, pj = ops.round_pairs(n_pairs)
        Z, rounds = ops.finish(n)
    finally:
        ops.close()
    return (Z, rounds) if return_rounds else Z

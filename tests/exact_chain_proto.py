"""Prototype (numpy, CPU) of an EXACT float32 column chain that does not serialise over the rows.

The reference forms its profile with ``np.mean(X, axis=0)`` / scipy's CSR mean (reference ``tl/_infercnv.py:385, :400``):
per column ONE sequential float32 chain ``s = fl(s + x_r)`` over the rows -- the shipped kernels (``k_colchain``,
``k_colchain_csrq``) evaluate exactly that chain, so row shards on several GPUs have to take turns.

Inside one binade the chain is integer arithmetic.  With ``s = m u`` (``u = ulp(s) = 2^(e-23)``, ``2^23 <= m < 2^24``) and
``x >= 0``: ``fl(s + x) = (m + q) u`` with ``q = round_half_even(x / u)`` -- and ``q`` does not depend on ``m`` unless
``x / u`` has the fraction exactly 1/2 (a TIE: then the parity of ``m`` decides).  So for a block of rows whose chain stays
inside the binade it starts in, and that holds no tie, the block's whole effect is ``m += Q`` with ``Q = sum of q`` --
integer additions, any order, any rank, any time: the block only needs to know the BINADE of its start value, which a
float64 estimate of the prefix sum gives.  A block is valid iff (exact test, the data are non-negative so ``m`` only
grows): the assumed binade is the binade of the true start, no tie, ``m + Q < 2^24``.  Invalid blocks -- the first rows of
a column (the sum doubles with every few entries), one block per later binade crossing, the rare tie, a start within the
estimate's error of a power of two -- are replayed sequentially, in order, which is what the chain kernels do.

``chain_by_blocks`` restates that in numpy (one matrix, the rows in ``block`` -row blocks) and counts the replayed blocks by
cause; ``tests/test_exact_chain_proto.py`` holds it ``array_equal`` to numpy / scipy on the means fuzzer's generator and on
adversarial inputs, ``rank_records`` / ``rank_scan`` are the two halves a rank of a row-sharded job would run
(``tests/test_dist_gloo.py``), and ``tools/exact_chain_stats.py`` prints the replay volume at config 3's / config 4's
geometry (profiles/r06_exact_chain_prototype.txt).  Test infrastructure: nothing in the product imports it.
"""
from __future__ import annotations

import numpy as np

TWO24 = float(1 << 24)
_STAT_KEYS = ("blocks", "replayed", "no_start", "estimate", "crossing", "tie_or_sign", "ties")


def exponent_f32(s):
    """e with 2^e <= s < 2^(e+1) for positive NORMAL float32 values, INT_MIN elsewhere (zero, subnormal, negative, inf, nan)."""
    s = np.asarray(s, dtype=np.float32)
    m, e = np.frexp(s.astype(np.float64))  # s = m 2^e, 0.5 <= m < 1
    ok = np.isfinite(s) & (s >= np.float32(2.0 ** -126))
    return np.where(ok, e - 1, np.iinfo(np.int32).min).astype(np.int64)


def block_records(xb, e_assumed):
    """One block of rows (``xb``: rows x columns, float32, absent entries 0) under the assumed binades ``e_assumed``
    (per column; INT_MIN = none): ``Q`` = sum of the rounded quotients (float64 holding an exact integer), ``bad`` = the
    block must be replayed whatever its start (a tie, a negative / non-finite entry, no binade, Q alone leaves the binade)."""
    e = np.asarray(e_assumed, dtype=np.int64)
    known = e > np.iinfo(np.int32).min
    u = np.ldexp(1.0, np.where(known, e, 0) - 23)
    with np.errstate(over="ignore", invalid="ignore"):
        t = xb.astype(np.float64) / u  # exact: a power of two
        k = np.floor(t)
        f = t - k
        q = k + (f > 0.5)
        tie = (f == 0.5).any(axis=0)
        neg = ((xb < 0) | ~np.isfinite(xb)).any(axis=0)
        Q = q.sum(axis=0)
    big = ~(Q < TWO24)
    return np.where(known & ~big & ~neg, Q, 0.0), ~known | tie | neg | big, tie & known & ~neg


def _replay(s, xb, cols):
    """The sequential float32 chain over the rows of ``xb`` for the columns ``cols`` (numpy's own adds, row by row)."""
    acc = s[cols].copy()
    for r in range(xb.shape[0]):
        acc = (acc + xb[r, cols]).astype(np.float32)
    s[cols] = acc


def scan_block(s, xb, e_assumed, Q, bad, stats=None):
    """Apply one block to the exact chain values ``s`` (float32, in place): ``m += Q`` where the block is valid for the
    TRUE start, sequential replay elsewhere."""
    e_s = exponent_f32(s)
    same = e_s == e_assumed
    ok = (~bad) & same
    u = np.ldexp(1.0, np.where(ok, e_s, 0) - 23)
    m = s.astype(np.float64) / u
    stays = (m + Q) < TWO24
    new = ((m + Q) * u).astype(np.float32)  # exact: an integer below 2^24 times a power of two
    if stats is not None:
        nz = xb.any(axis=0)  # (a block without entries in a column costs nothing to replay)
        none = e_s == np.iinfo(np.int32).min
        stats["blocks"] += int(nz.sum())
        stats["replayed"] += int((nz & ~(ok & stays)).sum())
        stats["no_start"] += int((nz & none).sum())                      # the chain has not reached a normal value yet
        stats["estimate"] += int((nz & ~none & ~same).sum())             # the estimate's binade is not the start's
        stats["crossing"] += int((nz & ok & ~stays).sum())               # the chain leaves its binade inside the block
        stats["tie_or_sign"] += int((nz & ~none & same & bad).sum())     # a tie / a negative or non-finite entry
    ok &= stays
    redo = np.flatnonzero(~ok)
    s[ok] = new[ok]
    if redo.size:
        _replay(s, xb, redo)


def chain_by_blocks(X, start=None, block=64, estimate_start=None, stats=None):
    """Float32 column chain of the rows of ``X`` (rows x columns float32 ndarray) continued from ``start``: equal,
    bit for bit, to ``for r: s = float32(s + X[r])``.  ``estimate_start`` (float64 per column; None: ``start``) is what
    the blocks take their binades from -- in a sharded job the all-gathered float64 sums of the earlier ranks."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, c = X.shape
    s = np.zeros(c, dtype=np.float32) if start is None else np.array(start, dtype=np.float32, copy=True)
    est = s.astype(np.float64) if estimate_start is None else np.array(estimate_start, dtype=np.float64, copy=True)
    if stats is not None:
        for k in _STAT_KEYS:
            stats.setdefault(k, 0)
    # pass A: float64 block sums -> the estimate of every block's start; pass B: the records; then the scan.  (Both
    # passes are independent per block: that is the point.)
    for r0 in range(0, n, block):
        xb = X[r0:r0 + block]
        e_assumed = exponent_f32(est.astype(np.float32))
        Q, bad, tie = block_records(xb, e_assumed)
        if stats is not None:
            stats["ties"] += int(tie.sum())
        scan_block(s, xb, e_assumed, Q, bad, stats)
        with np.errstate(invalid="ignore", over="ignore"):
            est = est + xb.sum(axis=0, dtype=np.float64)
    return s


def csr_scaled_dense(X_csr, count):
    """scipy's CSR mean adds ``fl32(x * fl32(1 / n))`` of the STORED entries row by row (reference :385 through
    ``(X * (1 / n)).sum(axis=0)``): the same chain over the scaled entries with zeros where nothing is stored (adding a
    zero to a non-negative sum changes nothing)."""
    Xs = X_csr.tocsr().astype(np.float32).copy()
    Xs.data = (Xs.data * np.float32(1.0 / count)).astype(np.float32)
    return Xs


# ---- the two halves of a rank in a row-sharded job ------------------------------------------------------------------
def rank_records(X_local, est_start, block=64):
    """Concurrent on every rank: the records of this rank's blocks from the float64 estimate of its start (the
    all-gathered totals of the earlier ranks): ``(e_assumed, Q, bad)`` per block, and the rank's float64 column totals."""
    X_local = np.ascontiguousarray(X_local, dtype=np.float32)
    est = np.array(est_start, dtype=np.float64, copy=True)
    recs = []
    for r0 in range(0, X_local.shape[0], block):
        xb = X_local[r0:r0 + block]
        e_assumed = exponent_f32(est.astype(np.float32))
        Q, bad, _ = block_records(xb, e_assumed)
        recs.append((e_assumed, Q, bad))
        est = est + xb.sum(axis=0, dtype=np.float64)
    return recs


def rank_scan(X_local, s_start, recs, block=64, stats=None):
    """Sequential over the ranks, but only a scan over the records (and the replays): the exact chain values after this
    rank's rows, from the exact values handed over by the previous rank."""
    X_local = np.ascontiguousarray(X_local, dtype=np.float32)
    s = np.array(s_start, dtype=np.float32, copy=True)
    if stats is not None:
        for k in _STAT_KEYS:
            stats.setdefault(k, 0)
    for b, (e_assumed, Q, bad) in enumerate(recs):
        scan_block(s, X_local[b * block:(b + 1) * block], e_assumed, Q, bad, stats)
    return s


def sharded_chain(X_local, dist, block=64, group=None):
    """The row-sharded form over a torch.distributed group (prototype of the exchange, numpy compute): every rank adds
    its float64 column totals (concurrent), ONE all-gather makes every rank's start estimate, every rank forms its
    block records (concurrent), then the exact chain values travel rank to rank -- a scan over the records and the
    replays, not a pass over the rows -- and the last rank broadcasts.  Returns the chain values after ALL rows, identical
    on every rank and bit-equal to the sequential chain over the whole matrix."""
    import torch

    rank, size = dist.get_rank(group), dist.get_world_size(group)
    X_local = np.ascontiguousarray(X_local, dtype=np.float32)
    totals = torch.from_numpy(X_local.sum(axis=0, dtype=np.float64))
    gathered = [torch.empty_like(totals) for _ in range(size)]
    dist.all_gather(gathered, totals, group=group)
    est = np.sum([g.numpy() for g in gathered[:rank]], axis=0) if rank else np.zeros(X_local.shape[1])
    recs = rank_records(X_local, est, block=block)
    s = torch.zeros(X_local.shape[1], dtype=torch.float32)
    if rank > 0:
        dist.recv(s, rank - 1, group=group)
    stats = {}
    out = torch.from_numpy(rank_scan(X_local, s.numpy(), recs, block=block, stats=stats))
    if rank < size - 1:
        dist.send(out, rank + 1, group=group)
    dist.broadcast(out, size - 1, group=group)
    return out.numpy(), stats

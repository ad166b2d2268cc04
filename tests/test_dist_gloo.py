"""CPU tests of the multi-GPU host logic with the gloo backend, world_size = 2.

The collectives and the shard/chunk bookkeeping of ``infercnvpy_amd.dist`` are device-agnostic;
here the per-rank GPU results (column sums, per-cell moments) are stood in by numpy computed from
the oracle, so the N > 1 path -- shard bounds, reference-mean all-reduce, chunk-moment all-reduce
for unaligned shards -- is checked end to end against the single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from infercnvpy_amd import dist as icd
from oracle import infercnv_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, align, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_obs, chunksize = 230, 50
        v = cases.synthetic_var([300, 120, 101, 60])
        X = cases.synthetic_expr(n_obs, len(v["names"]), seed=31)
        labels = np.array(["n1"] * 40 + ["n2"] * 30 + ["t"] * 160)[np.random.RandomState(1).permutation(n_obs)]
        r0, r1 = icd.shard_bounds(n_obs, world, chunksize, align=align)[rank]
        Xl, ll = X[r0:r1], labels[r0:r1]

        # (1) reference means: local float64 sums (stand-in for icv_colsum) -> one all-reduce
        cats = ["n1", "n2"]
        sums = torch.from_numpy(np.vstack([Xl[ll == c].sum(axis=0, dtype=np.float64) for c in cats]))
        counts = [int((ll == c).sum()) for c in cats]
        ref = icd.reference_means(sums, counts, np.float32)
        exp = np.vstack([X[labels == c].sum(axis=0, dtype=np.float64) / (labels == c).sum() for c in cats])
        np.testing.assert_array_equal(ref, exp.astype(np.float32))

        # (2) thresholds: per-cell moments of the un-thresholded x_res (stand-in for the kernel output)
        _, x_res, _, _ = O.infercnv_chunk(Xl, v["chromosome"], v["start"], ref, 3, 100, 10, None)
        stats = torch.from_numpy(np.stack([x_res.sum(axis=1), (x_res ** 2).sum(axis=1)], axis=1))
        thr = icd.global_thresholds(stats, r0, n_obs, chunksize, x_res.shape[1], 1.5)
        # single-process truth: the oracle's per-chunk thresholds on the full matrix
        _, _, _, thr_exp = O.infercnv(X, v["chromosome"], v["start"], reference=ref, chunksize=chunksize,
                                      exclude_chromosomes=None)
        np.testing.assert_allclose(thr.numpy(), np.array(thr_exp), rtol=1e-12)

        # (3) applying them shard by shard reproduces the oracle's thresholded rows
        out = x_res.copy()
        for i in range(out.shape[0]):
            t = float(thr[(r0 + i) // chunksize])
            out[i][np.abs(out[i]) < t] = 0
        _, full, _, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref, chunksize=chunksize,
                                   exclude_chromosomes=None)
        np.testing.assert_array_equal(out, full.toarray()[r0:r1])
        q.put((rank, "ok", (r0, r1)))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("align", [True, False])
def test_two_rank_reference_and_thresholds(align):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, align, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, _ in results:
        assert status == "ok", f"rank {rank}: {status}"
    bounds = sorted(b for _, _, b in results)
    assert bounds[0][0] == 0 and bounds[0][1] == bounds[1][0] and bounds[1][1] == 230
    if align:
        assert bounds[0][1] % 50 == 0


def test_shard_bounds_properties():
    for n, w, cs in [(1_000_000, 8, 5000), (100_000, 8, 5000), (12_345, 4, 5000), (4, 8, 2), (0, 2, 10)]:
        b = icd.shard_bounds(n, w, cs)
        assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
        for (a0, a1), (b0, b1) in zip(b, b[1:]):
            assert a1 == b0 and a0 <= a1
        for r0, r1 in b:
            assert r0 % cs == 0 or r0 == n
        sizes = [r1 - r0 for r0, r1 in b]
        assert max(sizes) - min(sizes) <= cs
    assert icd.shard_bounds(1_000_000, 8, 5000) == [(i * 125_000, (i + 1) * 125_000) for i in range(8)]
    assert icd.shard_bounds(10, 3, 5, align=False) == [(0, 4), (4, 7), (7, 10)]


def _agree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # shard_bounds(12500, 5, 5000, align=False)-style partitions mix aligned and unaligned
        # shards; every rank must take the same (collective) branch.  Scaled down: 3 ranks, chunksize 50.
        n_obs, cs = 150, 50
        bounds = [(0, 100), (100, 130), (130, 150)]
        r0, r1 = bounds[rank]
        local = icd.shards_aligned([(r0, r1)], n_obs, cs)
        assert local == (rank == 0)
        assert icd.agree_aligned(local) is False
        assert icd.shards_aligned(bounds, n_obs, cs) is False
        ok_bounds = icd.shard_bounds(n_obs, world, cs)
        assert icd.agree_aligned(icd.shards_aligned([ok_bounds[rank]], n_obs, cs)) is True
        q.put((rank, "ok", None))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_three_ranks_agree_on_the_threshold_branch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, _ in results:
        assert status == "ok", f"rank {rank}: {status}"


def test_shards_aligned():
    assert icd.shards_aligned(icd.shard_bounds(1_000_000, 8, 5000), 1_000_000, 5000)
    assert icd.shards_aligned(icd.shard_bounds(12_345, 4, 5000), 12_345, 5000)
    assert icd.shards_aligned([(0, 10), (10, 10), (10, 17)], 17, 5)
    assert not icd.shards_aligned(icd.shard_bounds(12_500, 5, 5000, align=False), 12_500, 5000)
    assert not icd.shards_aligned([(0, 7), (7, 10)], 10, 5)


def test_chunk_moments_segments():
    rng = np.random.RandomState(0)
    stats = torch.from_numpy(rng.rand(137, 2))
    for row0, cs in [(0, 50), (30, 50), (49, 50), (100, 25), (7, 200)]:
        n_chunks = (row0 + 137 + cs - 1) // cs
        m = icd.chunk_moments(stats, row0, cs, n_chunks).numpy()
        for k in range(n_chunks):
            lo, hi = max(row0, k * cs) - row0, min(row0 + 137, (k + 1) * cs) - row0
            if hi <= lo:
                assert (m[k] == 0).all()
                continue
            assert m[k, 0] == hi - lo
            np.testing.assert_allclose(m[k, 1:], stats[lo:hi].sum(dim=0).numpy(), rtol=1e-14)


# --------------------------------------------------------------------------- #
# config 5: row-block distances on every rank -> gathered on rank 0 -> Ward -> broadcast
# --------------------------------------------------------------------------- #
def _cpu_distance_rows(x_all, r0, r1, out):
    x = x_all.numpy().astype(np.float64)
    d = ((x[r0:r1, None, :] - x[None, :, :]) ** 2).sum(-1)
    out[: r1 - r0, : x.shape[0]] = torch.from_numpy(d.astype(np.float32))
    return out


def _cpu_ward(dist_sq):
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import squareform

    d = np.sqrt(dist_sq.numpy().astype(np.float64))
    return linkage(squareform((d + d.T) / 2, checks=False), method="ward")


def _ward_worker(rank, world, port, ragged, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(8)
        n = 61 if ragged else 60
        X = (rng.standard_normal((n, 12)) + 3 * rng.randint(0, 3, (n, 1))).astype(np.float32)
        bounds = [(0, 25), (25, n)] if ragged else [(0, 30), (30, 60)]
        r0, r1 = bounds[rank]
        Z = icd.ward_linkage_sharded(torch.from_numpy(X[r0:r1]), distance_rows=_cpu_distance_rows, ward=_cpu_ward)
        Zs = O.ward_linkage(X)
        np.testing.assert_allclose(Z[:, 2], Zs[:, 2], rtol=1e-5)
        np.testing.assert_array_equal(Z[:, [0, 1, 3]], Zs[:, [0, 1, 3]])
        q.put((rank, "ok", None))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_sharded_ward_linkage(ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ward_worker, args=(r, 2, port, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, _ in results:
        assert status == "ok", f"rank {rank}: {status}"

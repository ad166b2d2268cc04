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
# config 5: sharded distance tiles + sharded Ward rounds.  The step kernels are replaced by a numpy
# restatement with the same interface (icd.HipWardSteps) and small super-rows, so that the exchanges of
# ward_linkage_sharded (mirror blocks, partner rows, new columns, neighbour all-reduce) run on CPU.
# --------------------------------------------------------------------------- #
class NumpyWardSteps:
    def __init__(self, n, layout):
        self.n, self.L = n, layout
        self.alive = np.ones(n, bool)
        self.cstate = np.full(n, -1, np.int64)   # -2 dead, -1 unchanged, >= 0 absorbed slot
        self.size_old = np.ones(n, np.int64)
        self.size_new = np.ones(n, np.int64)
        self.nn = np.zeros(n, np.int64)
        self.dmin = np.zeros(n, np.float32)
        self.pair_d = np.zeros(n, np.float32)
        self.act = np.arange(n)
        self.merges = []      # (i, j, d, size) in log order
        self.last = []        # the previous round's (i, j, size_i, size_j, d)
        self.rounds = 0

    def close(self):
        pass

    def _mine(self, r):
        return self.L.row_owner([r])[0] == self.L.rank

    def _row(self, d_local, r):
        return d_local[int(self.L.lrow([r])[0])]

    def distances(self, x_all, d_local, mirror):
        L, S, T = self.L, self.L.S, self.L.S // 8
        x = x_all.numpy().astype(np.float64)
        n = x.shape[0]
        row0, col0, dir_off, mir_off = L.tile_plan()
        dl, mr = d_local.numpy().reshape(-1), mirror.numpy().reshape(-1)
        for r0, c0, do, mo in zip(row0, col0, dir_off, mir_off):
            for ty in range(r0, min(r0 + S, n), T):
                for tx in range(c0, min(c0 + S, n), T):
                    if tx < ty:
                        continue
                    rr, cc = np.arange(ty, min(ty + T, n)), np.arange(tx, min(tx + T, n))
                    blk = ((x[rr][:, None, :] - x[cc][None, :, :]) ** 2).sum(-1).astype(np.float32)
                    dl[(do + (rr[:, None] - r0) * L.ld + (cc[None, :] - c0)).ravel()] = blk.ravel()
                    if tx > ty:
                        mr[(mo + (cc[None, :] - c0) * L.rows_padded + (rr[:, None] - r0)).ravel()] = blk.ravel()

    @staticmethod
    def _lw(dac, dbc, dab, na, nb, nc):
        v = ((na + nc) * np.float64(dac) + (nb + nc) * np.float64(dbc) - nc * np.float64(dab)) / (na + nb + nc)
        return np.maximum(v, 0).astype(np.float32)

    def merge(self, d_local, stage, pslot):
        D = d_local.numpy()
        st = stage.numpy() if stage is not None else None
        cs, so, sn = self.cstate, self.size_old, self.size_new
        for p, (i, j, si, sj, pd) in enumerate(self.last):
            if not self._mine(i):
                continue
            Dr = self._row(D, i)
            Dj = st[pslot[p]] if pslot[p] >= 0 else self._row(D, j)
            new = Dr.copy()
            best, best_c = np.inf, -1
            for c in np.flatnonzero(cs != -2):
                if c == i:
                    continue
                if cs[c] == -1:
                    v = self._lw(Dr[c], Dj[c], pd, si, sj, so[c])
                else:
                    cl = cs[c]
                    xk = self._lw(Dr[c], Dj[c], pd, si, sj, so[c])
                    xl = self._lw(Dr[cl], Dj[cl], pd, si, sj, so[cl])
                    v = self._lw(xk, xl, self.pair_d[c], so[c], so[cl], si + sj)
                new[c] = v
                if v < best:
                    best, best_c = v, c
            Dr[:] = new
            self.nn[i], self.dmin[i] = best_c, best

    def gather(self, d_local, rows_t, slots_t, out):
        # in-place column layout: the column of a cluster is its slot
        out.copy_(d_local[rows_t][:, slots_t.long()])

    def scatter(self, d_local, v, vrow_p):
        D, V = d_local.numpy(), v.numpy()
        vrow_i = np.asarray([self.last[p][0] for p in vrow_p])  # slot of the merge every row of v belongs to
        for lr in range(self.L.rows_padded):
            c = int(self.L.supers_of[self.L.rank][lr // self.L.S]) * self.L.S + lr % self.L.S
            if c < self.n and self.cstate[c] == -1:
                D[lr, vrow_i] = V[:, lr]

    def scan(self, d_local):
        D = d_local.numpy()
        for r in self.act:
            if not self._mine(r):
                continue
            row = self._row(D, r)[: self.n].copy()
            row[~self.alive] = np.inf
            row[r] = np.inf
            c = int(np.argmin(row))  # first minimum: lowest index on ties
            self.nn[r], self.dmin[r] = c, row[c]

    def _listed(self):
        return np.concatenate([np.asarray([m[0] for m in self.last], dtype=np.int64), np.asarray(self.act, dtype=np.int64)])

    def pack(self, k, device):
        rows = self._listed()
        mine = self.L.row_owner(rows) == self.L.rank if len(rows) else np.zeros(0, bool)
        nn = torch.from_numpy(np.where(mine, self.nn[rows], 0).astype(np.int32))
        dm = torch.from_numpy(np.where(mine, self.dmin[rows], 0).astype(np.float32))
        assert len(rows) == k
        return nn, dm

    def unpack(self, nn, dm):
        rows = self._listed()
        self.nn[rows] = nn.numpy()
        self.dmin[rows] = dm.numpy()

    def pairs(self, d_local, all_active):
        self.size_old[:] = self.size_new
        self.cstate[:] = np.where(self.alive, -1, -2)
        live = np.flatnonzero(self.alive)
        self.last = []
        for r in live:
            c = self.nn[r]
            if c > r and self.nn[c] == r:
                self.last.append((int(r), int(c), int(self.size_old[r]), int(self.size_old[c]), float(self.dmin[r])))
                self.merges.append((int(r), int(c), float(self.dmin[r]), int(self.size_old[r] + self.size_old[c])))
                self.cstate[r], self.cstate[c] = c, -2
                self.pair_d[r] = self.dmin[r]
                self.size_new[r] = self.size_old[r] + self.size_old[c]
                self.alive[c] = False
        self.act = [int(r) for r in live if self.cstate[r] == -1 and (all_active or self.cstate[self.nn[r]] != -1)]
        self.rounds += 1
        return int(self.alive.sum()), len(self.merges), len(self.last), len(self.act)

    def round_pairs(self, n_pairs):
        return (np.asarray([m[0] for m in self.last], dtype=np.int64), np.asarray([m[1] for m in self.last], dtype=np.int64))

    def finish(self, n):
        # merge log -> scipy linkage matrix (the host part of icv_ward_finish)
        m = len(self.merges)
        height, slot_h = np.zeros(m), np.zeros(n)
        for p, (i, j, d, _) in enumerate(self.merges):
            height[p] = max(np.sqrt(d), slot_h[i], slot_h[j])
            slot_h[i] = height[p]
        order = np.argsort(height, kind="stable")
        cluster = np.arange(n)
        Z = np.zeros((m, 4))
        for q, p in enumerate(order):
            i, j, _, size = self.merges[p]
            a, b = cluster[i], cluster[j]
            Z[q] = [min(a, b), max(a, b), height[p], size]
            cluster[i] = n + q
        return Z, self.rounds


def _ward_worker(rank, world, port, n, ragged, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(8)
        X = (rng.standard_normal((n, 12)) + 3 * rng.randint(0, 3, (n, 1))).astype(np.float32)
        cut = np.linspace(0, n, world + 1).astype(int)
        if ragged:
            cut[1] -= 5
        r0, r1 = cut[rank], cut[rank + 1]
        Z, rounds = icd.ward_linkage_sharded(torch.from_numpy(X[r0:r1]), steps=NumpyWardSteps, super_rows=16,
                                             return_rounds=True)
        Zs = O.ward_linkage(X)
        assert 1 < rounds < n - 1
        np.testing.assert_allclose(Z[:, 2], Zs[:, 2], rtol=1e-5)
        np.testing.assert_array_equal(Z[:, [0, 1, 3]], Zs[:, [0, 1, 3]])
        q.put((rank, "ok", None))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,ragged", [(2, 60, False), (2, 61, True), (3, 150, False), (2, 10, False)])
def test_sharded_ward_rounds(world, n, ragged):
    """Sharded distance tiles + sharded Ward rounds (numpy step kernels, gloo): same tree as scipy.
    n = 10 with super-rows of 16: the second rank owns no rows at all."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ward_worker, args=(r, world, port, n, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, _ in results:
        assert status == "ok", f"rank {rank}: {status}"


def test_ward_layout_is_balanced_and_complete():
    """Folded cyclic ownership: every super-tile of the upper triangle is computed exactly once, the tile counts
    of the ranks differ by at most one row of super-tiles, and every rank holds about n / R rows."""
    for n, world in [(200_000, 8), (100_000, 4), (5000, 2), (1500, 3)]:
        seen = set()
        counts = []
        for r in range(world):
            L = icd.WardLayout(n, world, r)
            row0, col0, dir_off, mir_off = L.tile_plan()
            counts.append(len(row0))
            for a, b in zip(row0, col0):
                assert (a, b) not in seen and b >= a
                seen.add((int(a), int(b)))
            assert np.all(np.diff(dir_off) > 0)
        ns = -(-n // icd.SUPER)
        assert len(seen) == ns * (ns + 1) // 2
        assert max(counts) - min(counts) <= ns, (n, world, counts)
        assert sum(L.k) == ns and max(L.k) - min(L.k) <= 1


# ----------------------------------------------------------------------------------------------------------------------
# the reference-order chains handed from rank to rank, pipelined over column groups (dist.reference_means_chained)
# ----------------------------------------------------------------------------------------------------------------------
def test_chain_column_groups_cover_whole_lines():
    for n_cols, esz, t in [(20000, 4, 4), (20000, 4, 1), (20000, 8, 4), (1330, 4, 4), (33, 4, 4), (31, 4, 8), (5, 8, 3)]:
        g = icd.chain_column_groups(n_cols, esz, t)
        assert g[0][0] == 0 and g[-1][1] == n_cols and len(g) <= t
        for (a0, a1), (b0, b1) in zip(g, g[1:]):
            assert a1 == b0 and a0 < a1
        for c0, _ in g:
            assert c0 % (128 // esz) == 0  # groups start on 128-byte lines (the tile unit of k_colchain)
    assert icd.chain_column_groups(20000, 4, 4) == [(0, 4992), (4992, 9984), (9984, 14976), (14976, 20000)]


class _FakeShard:
    """Stands in for _engine.DeviceMatrix on the CPU: the rows of one rank as a torch tensor."""

    def __init__(self, x):
        self.x = torch.from_numpy(np.ascontiguousarray(x))
        self.shape = tuple(x.shape)
        self.dtype = self.x.dtype
        self.device = torch.device("cpu")
        self.format = 0  # _lib.ICV_DENSE


def _cpu_column_chain(dm, acc, rows, count, row0=0, row1=None, cols=None):
    """icv_colchain restated: one sequential chain per column in the matrix dtype, rows ascending."""
    c0, c1 = (0, dm.shape[1]) if cols is None else cols
    a = acc.numpy()
    x = dm.x.numpy()
    for r in (range(dm.shape[0]) if rows is None else rows):
        a[c0:c1] = a[c0:c1] + x[r, c0:c1]
    return acc


def _chain_worker(rank, world, port, members, t_groups, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infercnvpy_amd import _engine

        _engine.column_chain = _cpu_column_chain
        _engine.chain_mean = lambda acc, count, is_csr: acc / float(count)  # (float32 tensor / python float: float32)
        group = None if members is None else dist.new_group(ranks=members)
        order = list(range(world)) if members is None else sorted(members)  # (new_group sorts its ranks)
        n_obs, n_genes = 1000, 333
        X = cases.synthetic_expr(n_obs, n_genes, seed=5)
        labels = np.array(["a", "b", "c"])[np.random.RandomState(2).randint(0, 3, n_obs)]
        bounds = icd.shard_bounds(n_obs, len(order), 100)
        if rank in order:
            r0, r1 = bounds[order.index(rank)]
            dm = _FakeShard(X[r0:r1])
            ll = labels[r0:r1]
            means = icd.reference_means_chained(dm, [n_obs], group=group, n_col_groups=t_groups).numpy()
            np.testing.assert_array_equal(means[0], X.mean(axis=0))  # numpy's own order: the whole matrix at once
            cats = ["a", "b"]
            means = icd.reference_means_chained(dm, [int((labels == c).sum()) for c in cats],
                                                [np.nonzero(ll == c)[0] for c in cats], group=group,
                                                n_col_groups=t_groups).numpy()
            for gi, c in enumerate(cats):
                np.testing.assert_array_equal(means[gi], X[labels == c].mean(axis=0))
        q.put((rank, "ok", None))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,members,t_groups", [(2, None, 1), (3, None, 4), (3, [2, 0], 3)])
def test_chained_means_over_ranks_are_numpys_bits(world, members, t_groups):
    """Every rank ends up with np.mean(X, axis=0) of the WHOLE matrix bit for bit: the float32 chains continue from
    rank to rank, pipelined over column groups; with a subgroup the peers are the group's members in group order
    (the default group's ranks were used)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_worker, args=(r, world, port, members, t_groups, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, _ in results:
        assert status == "ok", f"rank {rank}: {status}"


def _exact_blocks_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import exact_chain_proto as P

        # config 3's sharding: row shards aligned to 5000-cell chunks, config 2's generator (4 000 of its columns: the
        # columns are independent, and the CPU suite has to stay short)
        n_obs, n_genes = 5000 * world, 4000
        bounds = icd.shard_bounds(n_obs, world, 5000)
        r0, r1 = bounds[rank]
        X_local = np.vstack([cases.synthetic_expr(5000, n_genes, seed=300 + k) for k in range(r0 // 5000, r1 // 5000)])
        got, stats = P.sharded_chain(X_local, dist, block=64)
        # the chained form (what dist.reference_means_chained evaluates): one sequential chain over ALL rows
        # (numpy adds a C-contiguous matrix row by row in float32: np.add.reduce IS that chain, tests/test_exact_chain_proto.py)
        want = np.add.reduce(np.vstack([cases.synthetic_expr(5000, n_genes, seed=300 + k) for k in range(n_obs // 5000)]),
                             axis=0)
        assert want.dtype == np.float32
        np.testing.assert_array_equal(got.view(np.int32), want.view(np.int32))
        q.put((rank, "ok", stats))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exact_chain_by_blocks_over_ranks_is_the_chained_result(world):
    """the block-wise exact float32 chain (tests/exact_chain_proto.py) over row shards at config 3's
    geometry: every rank computes its block records CONCURRENTLY from the all-gathered float64 totals, only a scan travels
    rank to rank -- and the result is the chained means' bits (numpy's order over the whole matrix).  Ranks after the
    first replay a few per cent of their (block, column) pairs even here, 5000 rows into the chains (at config 3's
    125 000 rows per rank: 0.04-0.5 %, tools/exact_chain_stats.py -> profiles/r06_exact_chain_prototype.txt)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exact_blocks_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, stats in results:
        assert status == "ok", f"rank {rank}: {status}"
        if rank > 0:
            assert stats["replayed"] < 0.05 * stats["blocks"], stats

"""The hot path under a real process group on GPUs: 2 ranks (cuda:0, cuda:1), backend nccl (= RCCL over xGMI).

Skipped on boxes with fewer than 2 GPUs.  What runs on every rank is the product path: ``icv_colsum`` on the rank's
rows -> ``dist.reference_means`` (one all-reduce) -> ``dist.run_shard`` (chunk-aligned: no further collective;
unaligned: the chunk-moment all-reduce + ``icv_apply_threshold``).  The concatenated shards must equal the
single-GPU run bit for bit; the sharded Ward linkage must equal ``tl.ward_linkage`` on one GPU.

``test_sharded_ward_two_processes_one_gpu`` runs the sharded distance tiles and Ward rounds (the HIP step kernels
behind ``icv_pairwise_sqeuclidean_tiles`` / ``icv_ward_*``) with two processes on ONE GPU: the collectives go
through gloo and host copies there, everything else is the code path of the multi-GPU job.
``test_hot_path_ranks_on_one_gpu`` does the same for the headline path (BASELINE config 3: ``icv_colsum`` ->
``dist.reference_means`` -> ``dist.run_shard``), so that the sharded code runs under a process group on every
box, and ``test_public_api_shards_*`` run the multi-GPU ``tl.infercnv`` (``devices=``) with several shards on one
GPU (and on two GPUs where the box has them)."""
import os
import socket

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from infercnvpy_amd import _engine, dist as icd
        from infercnvpy_amd._plan import GenePlan

        v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
        plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
        n_obs, cs = 2300, 500
        X = torch.from_numpy(cases.synthetic_expr(n_obs, 20000, seed=51)).cuda()
        out = {}
        for align in (True, False):
            bounds = icd.shard_bounds(n_obs, world, cs, align=align)
            r0, r1 = bounds[rank]
            dm = _engine.DeviceMatrix(dense=X[r0:r1].contiguous())
            sums = _engine.column_sums(dm)
            ref = icd.reference_means(sums, [r1 - r0], "float32", device_out=True)[0].contiguous()
            # with and without the partition: both ways every rank must take the same branch
            for ab in (bounds, None):
                res = icd.run_shard(plan, dm, ref, global_row0=r0, n_obs_global=n_obs, chunksize=cs, all_bounds=ab)
                torch.cuda.synchronize()
                out[(align, ab is None)] = (r0, r1, res.out.cpu().numpy(), ref.cpu().numpy())
        # config 5: sharded distances + Ward against the single-GPU linkage
        Xc = torch.from_numpy(np.random.RandomState(3).standard_normal((2500, 64)).astype(np.float32)).cuda()
        b2 = icd.shard_bounds(2500, world, 1)
        Z = icd.ward_linkage_sharded(Xc[b2[rank][0]:b2[rank][1]].contiguous())
        q.put((rank, "ok", out, Z))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_single_gpu():
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, status, _, _ in results:
        assert status == "ok", f"rank {rank}: {status}"

    from infercnvpy_amd import _engine
    from infercnvpy_amd._plan import GenePlan
    from infercnvpy_amd.tl import ward_linkage

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    X = torch.from_numpy(cases.synthetic_expr(2300, 20000, seed=51)).cuda()
    dm = _engine.DeviceMatrix(dense=X)
    ref = (_engine.column_sums(dm)[0] / 2300).float()
    whole = _engine.run_hot_path(plan, dm, ref, chunksize=500).out.cpu().numpy()
    for key in results[0][2]:
        parts = [results[r][2][key] for r in range(2)]
        assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == 2300
        # the all-reduced float64 sums are added in another order than on one GPU: same float32 means
        np.testing.assert_array_equal(parts[0][3], parts[1][3])
        np.testing.assert_allclose(parts[0][3], ref.cpu().numpy(), rtol=1e-6)
        got = np.vstack([parts[0][2], parts[1][2]])
        if np.array_equal(parts[0][3], ref.cpu().numpy()):
            assert np.array_equal(got, whole), key
        else:  # a mean differs in its last bit: values agree to rounding, the zero pattern up to threshold ties
            assert np.mean((got == 0) != (whole == 0)) < 1e-4
    Xc = np.random.RandomState(3).standard_normal((2500, 64)).astype(np.float32)
    Z1 = ward_linkage(Xc)
    for r in range(2):
        np.testing.assert_array_equal(results[r][3], Z1)


def _ward_one_gpu_worker(rank, world, port, n, d, in_place, q):
    if in_place:
        os.environ["ICV_WARD_IN_PLACE"] = "1"  # the column layout without spare columns, on every rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infercnvpy_amd import dist as icd

        X = _ward_points(n, d)
        cut = np.linspace(0, n, world + 1).astype(int)
        Z, rounds = icd.ward_linkage_sharded(torch.from_numpy(X[cut[rank]:cut[rank + 1]]).cuda(), return_rounds=True)
        q.put((rank, "ok", Z, rounds))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


def _ward_points(n, d):
    rng = np.random.RandomState(n)
    return (rng.standard_normal((n, d)) * 0.4 + rng.standard_normal((7, d))[rng.randint(0, 7, n)]).astype(np.float32)


@pytest.mark.parametrize("world,n,d,in_place", [(2, 2500, 64, False), (3, 5300, 40, False), (2, 700, 16, False),
                                                  (2, 2500, 64, True)])
def test_sharded_ward_two_processes_one_gpu(world, n, d, in_place):
    """Sharded tiles + sharded rounds equal the one-GPU linkage bit for bit (n = 700: one super-row, the second
    rank holds nothing; n = 5300: six super-rows over three ranks, the last one partial; in_place: the step kernels
    of the layout without spare columns)."""
    import torch.multiprocessing as mp

    from infercnvpy_amd.tl import ward_linkage

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ward_one_gpu_worker, args=(r, world, port, n, d, in_place, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, status, _, _ in results:
        assert status == "ok", f"rank {rank}: {status}"
    Z1, rounds1 = ward_linkage(_ward_points(n, d), return_rounds=True)
    for r in range(world):
        assert results[r][3] == rounds1
        np.testing.assert_array_equal(results[r][2], Z1)


# ----------------------------------------------------------------------------------------------------------------------
# the hot path under a process group with every rank on cuda:0 (collectives: gloo through the host)
# ----------------------------------------------------------------------------------------------------------------------
_HP_GENES = [700, 320, 150, 100, 60]
# geometry -> (genes per chromosome, cells, chunksize); "cfg3" = BASELINE config 3's own geometry (20 000 genes on
# chr1..22, 5000-cell chunks: k_smooth_x16's chunk-moment partials cross a shard cut when the shards are unaligned)
_HP_GEOM = {"small": (_HP_GENES, 2300, 500), "cfg3": (cases.GENES_PER_CHROM_20K, 12_500, 5000),
            "tiny": (_HP_GENES, 1000, 500)}  # (two chunks: with three ranks one of them owns no rows)


def _hp_inputs(fmt, geom="small"):
    import scipy.sparse as sp

    genes, n, _ = _HP_GEOM[geom]
    v = cases.synthetic_var(genes, extra=(("chrX", 40), (None, 6)) if geom != "cfg3" else ())
    n_genes = len(v["names"]) - len(v["names"]) % 4
    for key in ("chromosome", "start"):
        v[key] = v[key][:n_genes]
    X = cases.synthetic_expr(n, n_genes, seed=51)
    if fmt == "csr":
        X[X < np.quantile(X, 0.9)] = 0
        X = sp.csr_matrix(X)
    labels = np.array(["n1", "n2", "t"])[np.random.RandomState(9).randint(0, 3, n)]
    return v, X, labels


def _hp_run(icd, _engine, plan, X, labels, rank, world, align, cats, window, geom, means):
    """One rank's share: column sums -> all-reduced means (or chained accumulators) -> run_shard.
    Returns (r0, r1, out, thr, ref)."""
    import torch

    _, n_all, cs = _HP_GEOM[geom]
    bounds = icd.shard_bounds(n_all, world, cs, align=align)
    r0, r1 = bounds[rank]
    dm = _engine.to_device_matrix(X[r0:r1], torch.float32)
    ll = labels[r0:r1]
    if means == "blocks" and cats is None and isinstance(X, np.ndarray):
        # the chain by integer blocks: the ranks' passes run concurrently, only a scan travels (dist.reference_means_blocks)
        st = {}
        ref = icd.reference_means_blocks(dm, n_all, n_col_groups=3, stats=st)
        # (a rank that continues running chains replays a few per cent of its blocks at this depth, far less at config 3's
        # 125 000 rows per rank: tests/test_gpu_refmean.py::test_chain_by_blocks_replays_little_at_config3_geometry)
        assert rank == 0 or r1 == r0 or geom != "cfg3" or st["replayed"] < 0.2 * st["blocks"], st
    elif means == "auto" and cats is None:
        ref = icd.reference_means_exact(dm, n_all)  # blocks from 4 ranks on (dense float32), else the chained form
    elif means in ("chain", "blocks", "auto"):
        if cats is None:
            ref = icd.reference_means_chained(dm, [n_all])
        else:
            ref = icd.reference_means_chained(dm, [int((labels == c).sum()) for c in cats],
                                              [np.nonzero(ll == c)[0] for c in cats])
    elif cats is None:
        sums = _engine.column_sums(dm) if r1 > r0 else torch.zeros((1, X.shape[1]), dtype=torch.float64, device="cuda")
        ref = icd.reference_means(sums, [r1 - r0], "float32", device_out=True).contiguous()
    else:
        grp = np.full(r1 - r0, -1, dtype=np.int32)
        for gi, c in enumerate(cats):
            grp[ll == c] = gi
        sums = _engine.column_sums(dm, grp, len(cats)) if r1 > r0 else \
            torch.zeros((len(cats), X.shape[1]), dtype=torch.float64, device="cuda")
        ref = icd.reference_means(sums, [int((ll == c).sum()) for c in cats], "float32", device_out=True).contiguous()
    ref_lo = ref[0].contiguous() if cats is None else ref.min(dim=0).values.contiguous()
    ref_hi = None if cats is None else ref.max(dim=0).values.contiguous()
    # with and without the partition: both ways every rank must take the same branch
    outs = {}
    for ab in (bounds, None):
        res = icd.run_shard(plan, dm, ref_lo, ref_hi, global_row0=r0, n_obs_global=n_all, chunksize=cs, all_bounds=ab)
        torch.cuda.synchronize()
        outs[ab is None] = (res.out.cpu().numpy(), None if res.thr is None else res.thr.cpu().numpy())
    # the same shard with the thresholds applied while X_cnv is packed to device CSR (what bench.py --gpus N times)
    if True:  # (an empty shard packs to an empty matrix)
        import scipy.sparse as sp

        _, pk = icd.run_shard(plan, dm, ref_lo, ref_hi, global_row0=r0, n_obs_global=n_all, chunksize=cs, all_bounds=bounds,
                              pack=True)
        got, exp = pk.to_scipy(), sp.csr_matrix(outs[False][0].astype(np.float64))
        assert np.array_equal(got.indptr, exp.indptr) and np.array_equal(got.indices, exp.indices)
        assert np.array_equal(got.data, exp.data)
    return r0, r1, outs, ref.cpu().numpy()


def _hp_worker(rank, world, port, fmt, window, geom, means, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infercnvpy_amd import _engine, dist as icd
        from infercnvpy_amd._plan import GenePlan

        v, X, labels = _hp_inputs(fmt, geom)
        plan = GenePlan(v["chromosome"], v["start"], window_size=window, step=10)
        out = {}
        for align in (True, False):
            for cats in (None, ["n1", "n2"]):
                out[(align, cats is None)] = _hp_run(icd, _engine, plan, X, labels, rank, world, align, cats, window,
                                                     geom, means)
        q.put((rank, "ok", out))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fmt,window,geom,means", [
    (2, "dense", 100, "small", "allreduce"), (3, "dense", 100, "small", "allreduce"), (2, "csr", 100, "small", "allreduce"),
    (3, "csr", 250, "small", "allreduce"), (3, "dense", 100, "small", "chain"), (2, "csr", 250, "small", "chain"),
    (3, "dense", 100, "cfg3", "allreduce"), (3, "dense", 100, "cfg3", "chain"), (2, "csr", 100, "cfg3", "chain"),
    (2, "dense", 100, "small", "blocks"), (3, "dense", 100, "cfg3", "blocks"), (3, "dense", 100, "tiny", "blocks"),
    (4, "dense", 100, "small", "auto"), (2, "csr", 100, "small", "auto")])
def test_hot_path_ranks_on_one_gpu(world, fmt, window, geom, means):
    """BASELINE config 3's code path (row shards, ONE all-reduce of the reference sums -- or the reference-order chains
    handed from rank to rank --, chunk-aligned and unaligned thresholds) with 2-3 ranks sharing cuda:0, at a small
    geometry and at config 3's own (20 000 genes, window 100, 5000-cell chunks).  means="chain": every rank holds the
    reference's own means (array_equal to the oracle = numpy / scipy) and the concatenated shards equal the
    single-process run bit for bit; means="allreduce": the same whenever the all-reduced means equal the single-process
    float64-sum means (they are rounded from float64 sums added in another order)."""
    import torch
    import torch.multiprocessing as mp

    from infercnvpy_amd import _engine
    from infercnvpy_amd._plan import GenePlan
    from oracle import infercnv_oracle as O

    _, n_all, cs = _HP_GEOM[geom]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hp_worker, args=(r, world, port, fmt, window, geom, means, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert all(status == "ok" for _, status, _ in results), "; ".join(f"rank {r}: {st}" for r, st, _ in results)

    v, X, labels = _hp_inputs(fmt, geom)
    plan = GenePlan(v["chromosome"], v["start"], window_size=window, step=10)
    dm = _engine.to_device_matrix(X, torch.float32)
    for (align, allmean), _ in results[0][2].items():
        cats = None if allmean else ["n1", "n2"]
        if means in ("chain", "blocks", "auto"):
            ref_np = np.asarray(O.reference_profile(X, labels if cats else None, cats, None, X.shape[1]))
            ref = torch.from_numpy(np.ascontiguousarray(ref_np)).cuda()
        elif allmean:
            ref = (_engine.column_sums(dm) / n_all).float()
        else:
            grp = np.full(n_all, -1, dtype=np.int32)
            for gi, c in enumerate(["n1", "n2"]):
                grp[labels == c] = gi
            cnt = torch.tensor([(labels == "n1").sum(), (labels == "n2").sum()], dtype=torch.float64, device="cuda")
            ref = (_engine.column_sums(dm, grp, 2) / cnt[:, None]).float()
        ref_lo = ref[0].contiguous() if allmean else ref.min(dim=0).values.contiguous()
        ref_hi = None if allmean else ref.max(dim=0).values.contiguous()
        whole = _engine.run_hot_path(plan, dm, ref_lo, ref_hi, chunksize=cs)
        w_out, w_thr = whole.out.cpu().numpy(), whole.thr.cpu().numpy()
        parts = [results[r][2][(align, allmean)] for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == n_all
        for a, b in zip(parts[:-1], parts[1:]):
            assert a[1] == b[0]
            np.testing.assert_array_equal(a[3], b[3])  # all ranks hold the same means
        if means in ("chain", "blocks", "auto"):  # the reference's own bits, on every rank
            np.testing.assert_array_equal(parts[0][3], ref.cpu().numpy())
        np.testing.assert_allclose(parts[0][3], ref.cpu().numpy(), rtol=1e-6, atol=1e-12)
        same_ref = np.array_equal(parts[0][3], ref.cpu().numpy())
        for no_bounds in (False, True):
            got = np.vstack([p[2][no_bounds][0] for p in parts])
            if same_ref and (align or fmt == "dense" or window == 100):
                assert np.array_equal(got, w_out), (align, allmean, no_bounds)
            elif same_ref:
                # unaligned shards of the long-window CSR kernel: thresholds from all-reduced per-cell moments
                # (another summation order than the per-chunk partials): ties at 1e-12 may flip
                assert np.mean((got == 0) != (w_out == 0)) < 1e-5
                np.testing.assert_allclose(got[(got != 0) & (w_out != 0)], w_out[(got != 0) & (w_out != 0)], atol=1e-6)
            else:  # a mean differs in its last bit: values agree to rounding, the zero pattern up to threshold ties
                assert np.mean((got == 0) != (w_out == 0)) < 1e-4
            # thresholds of the chunks a rank holds
            for p in parts:
                thr = p[2][no_bounds][1]
                k0 = p[0] // cs
                if p[1] > p[0]:
                    np.testing.assert_allclose(thr, w_thr[k0:k0 + len(thr)], rtol=1e-6 if not same_ref else 1e-11)


# ----------------------------------------------------------------------------------------------------------------------
# multi-GPU tl.infercnv: several row shards inside one call
# ----------------------------------------------------------------------------------------------------------------------
def _api_inputs(fmt, n_obs=1150):
    import pandas as pd
    import scipy.sparse as sp

    v = cases.synthetic_var(_HP_GENES, extra=(("chrX", 40), ("chrM", 5), (None, 6)))
    X = cases.synthetic_expr(n_obs, len(v["names"]), seed=77)
    labels = np.array(["n1", "n2", "t"])[np.random.RandomState(4).randint(0, 3, n_obs)]
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    obs = pd.DataFrame({"group": labels})
    if fmt == "csr":
        X = sp.csr_matrix(X)
    return X, obs, var


def _api_run(X, obs, var, **kw):
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    tm = {}
    out = cnv.tl.infercnv(SimpleAnnData(X, obs=obs, var=var), inplace=False, chunksize=100, _timings=tm, **kw)
    return out, tm


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_public_api_shards_on_one_gpu(fmt):
    """``tl.infercnv(devices=[0, 0, 0])``: three chunk-aligned row shards, each with its own uploader, plan, stream
    and CSR drain, on one GPU.  X_cnv is bit-equal to the one-shard result (also gene values), with ``reference=``
    given and with means formed from the matrix (shard k continues the column chains of shard k - 1)."""
    X, obs, var = _api_inputs(fmt)
    ref = np.asarray(X[:200].mean(axis=0), dtype=np.float64).ravel().astype(np.float32)
    for kw in (dict(reference=ref), dict(reference=ref, window_size=250), dict(reference=ref, calculate_gene_values=True),
               dict(reference=np.vstack([ref, ref * 0.5 + 0.01]).astype(np.float32))):
        (pos1, res1, gv1), tm1 = _api_run(X, obs, var, devices=[0], **kw)
        assert tm1["devices"] == [0] and "shards" not in tm1
        for devs in ([0, 0], [0, 0, 0]):
            (pos, res, gv), tm = _api_run(X, obs, var, devices=devs, **kw)
            assert tm["devices"] == devs and [s["rows"] for s in tm["shards"]] == ([600, 550] if len(devs) == 2 else [400, 400, 350])
            assert list(pos.items()) == list(pos1.items())
            assert res.shape == res1.shape and res.dtype == res1.dtype
            np.testing.assert_array_equal(res.indptr, res1.indptr)
            np.testing.assert_array_equal(res.indices, res1.indices)
            np.testing.assert_array_equal(res.data, res1.data)
            if gv1 is not None:
                np.testing.assert_array_equal(gv, gv1)
    for kw in (dict(), dict(reference_key="group", reference_cat=["n1", "n2"]), dict(reference_key="group", reference_cat="n1")):
        (_, res1, _), _ = _api_run(X, obs, var, devices=[0], **kw)
        (_, res3, _), tm = _api_run(X, obs, var, devices=[0, 0, 0], **kw)
        # the reference-order chains continue from shard to shard: the means, and X_cnv, are those of one shard
        np.testing.assert_array_equal(res3.indptr, res1.indptr)
        np.testing.assert_array_equal(res3.indices, res1.indices)
        np.testing.assert_array_equal(res3.data, res1.data)
        assert "reference_pass" in tm
    # n_jobs: the reference's knob.  More jobs than GPUs or chunks: capped; n_jobs=1: one shard
    (_, res_j, _), tm = _api_run(X, obs, var, n_jobs=64, reference=ref)
    import torch

    assert len(tm["devices"]) == min(torch.cuda.device_count(), 12)
    (_, res_1, _), tm = _api_run(X, obs, var, n_jobs=1, reference=ref)
    assert tm["devices"] == [torch.cuda.current_device()]
    np.testing.assert_array_equal(res_j.toarray(), res_1.toarray())


def test_public_api_shard_failure_is_raised_and_leaves_no_threads(monkeypatch):
    """A shard that fails (here: its second kernel launch) aborts the others, the exception reaches the caller and
    every helper thread (uploaders, drains, shard threads) is gone afterwards."""
    import threading

    from infercnvpy_amd import _engine

    X, obs, var = _api_inputs("dense")
    ref = np.asarray(X[:200].mean(axis=0), dtype=np.float64).ravel().astype(np.float32)
    real = _engine.run_hot_path
    calls = {"n": 0}
    lock = threading.Lock()

    def flaky(*a, **k):
        with lock:
            calls["n"] += 1
            n = calls["n"]
        if n == 2:
            raise RuntimeError("injected failure")
        return real(*a, **k)

    _api_run(X, obs, var, devices=[0], reference=ref)  # (the process-wide pinned staging ring starts its threads once)
    monkeypatch.setattr(_engine, "run_hot_path", flaky)
    before = threading.active_count()
    for devs in ([0, 0, 0], [0]):
        calls["n"] = 0 if len(devs) > 1 else 1
        with pytest.raises(RuntimeError, match="injected failure"):
            _api_run(X, obs, var, devices=devs, reference=ref)
        assert threading.active_count() == before
    monkeypatch.setattr(_engine, "run_hot_path", real)
    (_, res, _), _ = _api_run(X, obs, var, devices=[0, 0], reference=ref)  # and the next call works
    assert res.shape[0] == X.shape[0]


def test_public_api_shards_on_two_gpus():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    for fmt in ("dense", "csr"):
        X, obs, var = _api_inputs(fmt)
        ref = np.asarray(X[:200].mean(axis=0), dtype=np.float64).ravel().astype(np.float32)
        (_, res1, _), _ = _api_run(X, obs, var, devices=[0], reference=ref)
        (_, res2, _), tm = _api_run(X, obs, var, devices=[0, 1], reference=ref)
        assert [s["device"] for s in tm["shards"]] == [0, 1]
        np.testing.assert_array_equal(res1.toarray(), res2.toarray())
        (_, res3, _), _ = _api_run(X, obs, var, devices=[1], reference=ref)
        np.testing.assert_array_equal(res1.toarray(), res3.toarray())

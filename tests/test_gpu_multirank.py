"""The hot path under a real process group on GPUs: 2 ranks (cuda:0, cuda:1), backend nccl (= RCCL over xGMI).

Skipped on boxes with fewer than 2 GPUs.  What runs on every rank is the product path: ``icv_colsum`` on the rank's
rows -> ``dist.reference_means`` (one all-reduce) -> ``dist.run_shard`` (chunk-aligned: no further collective;
unaligned: the chunk-moment all-reduce + ``icv_apply_threshold``).  The concatenated shards must equal the
single-GPU run bit for bit; the sharded Ward linkage must equal ``tl.ward_linkage`` on one GPU.

``test_sharded_ward_two_processes_one_gpu`` runs the sharded distance tiles and Ward rounds (the HIP step kernels
behind ``icv_pairwise_sqeuclidean_tiles`` / ``icv_ward_*``) with two processes on ONE GPU: the collectives go
through gloo and host copies there, everything else is the code path of the multi-GPU job."""
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

"""Ad-hoc check: sharded linkage (2 processes on one GPU, gloo + host copies) against the one-GPU linkage at 30 000 cells."""
import os, sys, time, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, torch.multiprocessing as mp
from test_gpu_multirank import _ward_one_gpu_worker, _ward_points, _free_port
if __name__ == "__main__":
    n, d, world = 30000, 200, 2
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    t0 = time.time()
    procs = [ctx.Process(target=_ward_one_gpu_worker, args=(r, world, port, n, d, False, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=900) for _ in procs), key=lambda r: r[0])
    [p.join(60) for p in procs]
    print("sharded", [r[1][:60] for r in res], round(time.time() - t0, 1), "s")
    from infercnvpy_amd.tl import ward_linkage
    t0 = time.time(); Z1, r1 = ward_linkage(_ward_points(n, d), return_rounds=True); print("single", round(time.time() - t0, 2), "s", r1)
    for r in res:
        assert r[3] == r1 and np.array_equal(r[2], Z1)
    print("bit-equal", n)

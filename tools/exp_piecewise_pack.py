"""Developer experiment: does running threshold + CSR pack piece by piece right behind the smoothing of each piece keep
x_res in the Infinity Cache?  (GPU box)   python tools/exp_piecewise_pack.py [cells]
Config-2 geometry; per variant: HIP events around smoothing + pack of all pieces, median of 15."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]


def main():
    import numpy as np
    import torch

    import cases
    from infercnvpy_amd import _engine
    from infercnvpy_amd._plan import GenePlan

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K, seed_perm=1)
    G = len(v["names"])
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.empty((n, G), dtype=torch.float32, device="cuda")
    X.copy_(torch._standard_gamma(torch.full((n, G), 0.3, device="cuda"), generator=g))
    X[torch.rand((n, G), device="cuda", generator=g) > 0.19] = 0
    dm = _engine.DeviceMatrix(dense=X)
    ref = X.mean(0)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    cs = 5000

    def run(piece):
        outs = []
        for r0 in range(0, n, piece):
            r1 = min(n, r0 + piece)
            res = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=1.5, chunksize=cs, apply=False, row0=r0, row1=r1)
            outs.append(_engine.threshold_csr(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=cs, row0=r0, row1=r1))
        return outs

    def smooth_only(piece):
        for r0 in range(0, n, piece):
            _engine.run_hot_path(plan, dm, ref, dynamic_threshold=1.5, chunksize=cs, apply=False, row0=r0, row1=min(n, r0 + piece))

    for fn, name in ((run, "smooth+pack"), (smooth_only, "smooth only")):
        for piece in (n, n // 2, n // 4, n // 10, n // 20):
            for _ in range(3):
                fn(piece)
            torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn(piece)
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b))
            print(f"{name:12s} piece {piece:7d} rows ({piece * plan.n_windows * 4 / 1e6:6.0f} MB of x_res): median {np.median(ts):.3f} ms  min {np.min(ts):.3f}", flush=True)


if __name__ == "__main__":
    main()

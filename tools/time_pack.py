"""Developer timing of the threshold + CSR pack stage, kernel by kernel (GPU box):
    python tools/time_pack.py [cells] [reps]
Config-2 geometry (20 000 genes, window 100 / step 10, dense fp32); HIP events around each entry point."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]


def main():
    import numpy as np
    import torch

    import cases
    from infercnvpy_amd import _engine, _lib
    from infercnvpy_amd._plan import GenePlan

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K, seed_perm=1)
    G = len(v["names"])
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.empty((n, G), dtype=torch.float32, device="cuda")
    X.copy_(torch._standard_gamma(torch.full((n, G), 0.3, device="cuda"), generator=g))
    X[torch.rand((n, G), device="cuda", generator=g) > 0.19] = 0
    dm = _engine.DeviceMatrix(dense=X)
    ref = X.mean(0)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    res = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=1.5, chunksize=5000, apply=False)
    lib = _lib.load()
    W = plan.n_windows

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts)), float(np.min(ts))

    out = {}
    for tag, env in (("ring", None), ("per_row", "1")):
        if env:
            os.environ["ICV_NO_MASK_RING"] = env
        else:
            os.environ.pop("ICV_NO_MASK_RING", None)
        lib.icv_developer_knobs_reload()
        part = None

        def mask():
            nonlocal part
            part = _engine.threshold_mask(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=5000)

        out[f"mask[{tag}]"] = timed(mask)
        out[f"nnz[{tag}]"] = int(part.counts.sum().item())
    indptr = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    st = _engine._stream_ptr(torch)
    out["row_offsets"] = timed(lambda: _lib.check(lib.icv_row_offsets(_engine._ptr(part.counts), n, _engine._ptr(indptr), st)))
    indices = torch.empty(n * W, dtype=torch.int32, device="cuda")
    data = torch.empty(n * W, dtype=torch.float64, device="cuda")
    fill = lambda: _lib.check(lib.icv_csr_fill_masked(  # noqa: E731
        _engine._ptr(part.out), n, W, part.out.stride(0), _engine._ptr(part.mask), _engine._ptr(indptr),
        _engine._ptr(indices), _engine._ptr(data), st))
    os.environ["ICV_NO_FILL_RING"] = "1"
    lib.icv_developer_knobs_reload()
    out["fill[per_row]"] = timed(fill)
    nnz = int(indptr[-1].item())
    ref_i, ref_d = indices[:nnz].clone(), data[:nnz].clone()
    indices.zero_()
    data.zero_()
    os.environ.pop("ICV_NO_FILL_RING")
    lib.icv_developer_knobs_reload()
    out["fill[ring]"] = timed(fill)
    out["fill[ring] == fill[per_row]"] = bool(torch.equal(indices[:nnz], ref_i) and torch.equal(data[:nnz].view(torch.int64), ref_d.view(torch.int64)))
    os.environ.pop("ICV_NO_MASK_RING", None)
    lib.icv_developer_knobs_reload()
    out["threshold_csr"] = timed(lambda: _engine.threshold_csr(plan, dm, ref, None, res, lfc_clip=3.0, chunksize=5000))
    for k, val in out.items():
        print(k, val)


if __name__ == "__main__":
    main()

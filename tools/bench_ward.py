"""Timing of BASELINE config 5 on one MI355X: squared-Euclidean tile kernel (fp32 MFMA) + Ward rounds.

    python tools/bench_ward.py --cells 50000 --features 5000 [--scipy 4000]

Prints one JSON line per size: executed TFLOP/s of the distance kernel (n (n + 128) d flop: only tiles on or
above the diagonal are multiplied, the mirror image is a transposed copy),
seconds and round count of the Ward linkage; optionally scipy pdist+linkage on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, nargs="+", default=[20000])
    ap.add_argument("--features", type=int, default=5000)
    ap.add_argument("--clusters", type=int, default=30)
    ap.add_argument("--in-place", action="store_true", help="row stride n (no spare columns for the Ward rounds)")
    ap.add_argument("--scipy", type=int, default=0, help="also time scipy on this many cells (CPU, float64)")
    a = ap.parse_args()
    import torch
    from infercnvpy_amd import _engine

    for n in a.cells:
        g = torch.Generator(device="cuda").manual_seed(n)
        centres = torch.randn((a.clusters, a.features), device="cuda", generator=g) * 0.3
        lab = torch.randint(0, a.clusters, (n,), device="cuda", generator=g)
        x = centres[lab] + 0.2 * torch.randn((n, a.features), device="cuda", generator=g)
        _engine.pairwise_sqeuclidean(x[:256].contiguous())  # warm-up
        if a.in_place:  # row stride n: the Ward rounds update columns in place
            d2 = torch.empty((n, (n + 3) // 4 * 4), dtype=torch.float32, device="cuda")[:, :n]
        else:  # default allocation of the library wrapper: n / 2 spare columns when HBM allows
            d2 = torch.empty((n, (n + (n + 1) // 2 + 3) // 4 * 4), dtype=torch.float32, device="cuda")[:, :n]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _engine.pairwise_sqeuclidean(x, out=d2)
        torch.cuda.synchronize()
        t_pd = time.perf_counter() - t0
        t0 = time.perf_counter()
        Z, rounds = _engine.ward_linkage(d2, spare=not a.in_place)
        t_w = time.perf_counter() - t0
        rec = {"cells": n, "features": a.features, "pdist_s": round(t_pd, 4),
               # executed flops: tiles on / above the diagonal only
               "pdist_tflops_executed": round(1.0 * n * (n + 128) * a.features / t_pd / 1e12, 2),
               "ward_s": round(t_w, 4),
               "ward_rounds": rounds, "matrix_gb": round(4.0 * n * d2.stride(0) / 1e9, 2), "row_stride": d2.stride(0),
               "ward_matrix_passes_equiv_gbps": round(8.0 * n * n / t_w / 1e9, 1), "top_height": float(Z[-1, 2])}
        if a.scipy and n == a.cells[0]:
            from scipy.cluster.hierarchy import linkage
            from scipy.spatial.distance import pdist

            xs = x[: a.scipy].cpu().numpy().astype(np.float64)
            t0 = time.perf_counter()
            y = pdist(xs)
            t1 = time.perf_counter()
            linkage(y, method="ward")
            t2 = time.perf_counter()
            rec["scipy_cells"] = a.scipy
            rec["scipy_pdist_s"] = round(t1 - t0, 3)
            rec["scipy_ward_s"] = round(t2 - t1, 3)
        print(json.dumps(rec), flush=True)
        del d2, x
        torch.cuda.empty_cache()  # the library allocates its temporaries with hipMalloc, outside torch's cache


if __name__ == "__main__":
    main()

"""End-to-end timing of the public API (host AnnData-like in, CSR X_cnv on the host out) on one MI355X.

    python tools/bench_e2e.py --cells 200000 [--format dense|csr] [--density 0.07] [--window 100]

SURVEY.md 8(d): the device-resident rate (bench.py, `value`) and this PCIe-inclusive rate are both reported;
they differ by the H2D copy of the matrix (80 KB/cell dense) and the D2H copy of the packed result.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=200000)
    ap.add_argument("--format", default="dense", choices=["dense", "csr"])
    ap.add_argument("--density", type=float, default=0.07)
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    import pandas as pd
    import torch

    import cases
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    var = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    G = len(var["names"])
    # synthesise on the GPU (fast), bring to the host: the timed region starts from host memory
    g = torch.Generator(device="cuda").manual_seed(2)
    parts = []
    for r0 in range(0, a.cells, 50000):
        n = min(50000, a.cells - r0)
        x = torch._standard_gamma(torch.full((n, G), 0.3, device="cuda"), generator=g)
        x = torch.where(x < 0.5, torch.zeros_like(x), x)
        if a.format == "csr":
            keep = torch.rand((n, G), device="cuda", generator=g) < (a.density / 0.19)
            x = torch.where(keep, x, torch.zeros_like(x))
        parts.append(x.cpu().numpy())
        del x
    X = np.concatenate(parts) if len(parts) > 1 else parts[0]
    del parts
    if a.format == "csr":
        X = sp.csr_matrix(X)
    torch.cuda.empty_cache()
    vdf = pd.DataFrame({"chromosome": var["chromosome"], "start": var["start"], "end": var["end"]}, index=var["names"])
    ad = SimpleAnnData(X, obs=pd.DataFrame(index=[f"c{i}" for i in range(a.cells)]), var=vdf)
    best = None
    for _ in range(a.repeat):
        t0 = time.perf_counter()
        cnv.tl.infercnv(ad, window_size=a.window)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res = ad.obsm["X_cnv"]
    in_bytes = X.nbytes if a.format == "dense" else X.data.nbytes + X.indices.nbytes + X.indptr.nbytes
    print(json.dumps({"workload": f"tl.infercnv end to end, host {a.format} float32 {a.cells} x {G}, window {a.window}",
                      "seconds": round(best, 3), "cells_per_s": round(a.cells / best, 1),
                      "input_GB": round(in_bytes / 1e9, 2), "input_GBps": round(in_bytes / best / 1e9, 2),
                      "x_cnv_nnz_frac": round(res.nnz / (res.shape[0] * res.shape[1]), 4)}))


if __name__ == "__main__":
    main()

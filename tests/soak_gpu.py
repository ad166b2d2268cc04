"""Soak of the public call on the GPU box (test infrastructure; not collected by pytest):
    python tests/soak_gpu.py [n_calls]
`tl.infercnv` over and over -- dense / CSR input, one shard / two shards on the one GPU, with and without gene values,
now and then a call that fails on purpose (a reference category that does not exist) -- and what must NOT grow: the
process's threads, its resident memory, the HBM torch holds, pinned staging buffers.  Results are compared with the
first call of their kind (bit-identical: the path is deterministic)."""
import gc
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, _p)


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def main():
    import numpy as np
    import pandas as pd
    import scipy.sparse as sp
    import torch

    import cases
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    G = len(v["names"])
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    X = cases.synthetic_expr(12_000, G, seed=3)
    Xs = sp.csr_matrix(X)
    labels = np.random.RandomState(0).choice(["a", "b", "c"], size=X.shape[0])
    kinds = [
        ("dense", dict()), ("csr", dict()), ("dense 2 shards", dict(devices=[0, 0])), ("csr 2 shards", dict(devices=[0, 0])),
        ("csr w250 cat2", dict(window_size=250, reference_key="group", reference_cat=["a", "b"])),
        ("dense gene values", dict(calculate_gene_values=True, chunksize=2000)),
    ]
    first = {}
    t0 = time.time()
    stats = []
    for i in range(n_calls):
        name, kw = kinds[i % len(kinds)]
        ad = SimpleAnnData(Xs if "csr" in name else X, obs=pd.DataFrame({"group": labels}), var=var)
        if i % 17 == 16:
            try:
                cnv.tl.infercnv(ad, reference_key="group", reference_cat="no such category", **{k: w for k, w in kw.items() if k == "devices"})
                raise SystemExit("soak: the call with a missing category did not raise")
            except ValueError:
                pass
            continue
        _, res, gv = cnv.tl.infercnv(ad, inplace=False, **kw)
        key = (res.shape, int(res.nnz), float(res.data.sum()), None if gv is None else float(np.nansum(gv)))
        if name not in first:
            first[name] = (key, res.copy())
        else:
            assert key == first[name][0], (name, key, first[name][0])
            if i % 25 == 0:
                assert (res != first[name][1]).nnz == 0, name
        del res, gv, ad
        if i % 20 == 19 or i == n_calls - 1:
            gc.collect()
            stats.append((i + 1, threading.active_count(), rss_mb(), torch.cuda.memory_allocated() / 1e6,
                          torch.cuda.memory_reserved() / 1e6))
            print("calls %4d  threads %3d  rss %8.0f MB  hbm allocated %8.1f MB  reserved %8.0f MB" % stats[-1], flush=True)
    # warm = after every call shape has been seen a few times by torch's caching allocator, the HIP memory pool and glibc's
    # heap (their high-water marks settle within ~120 calls of the six shapes)
    warm = stats[5] if len(stats) > 11 else (stats[1] if len(stats) > 2 else stats[0])
    last = stats[-1]
    print(f"soak: {n_calls} calls in {time.time() - t0:.0f} s")
    ok = last[1] <= warm[1] and last[2] <= warm[2] * 1.10 + 200 and last[3] <= warm[3] + 64
    print("soak:", "ok" if ok else "GROWTH", f"(after {warm[0]} calls: {warm[1:]}; after {last[0]}: {last[1:]})")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

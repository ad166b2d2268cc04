"""Random cases of the reference-order column means on the GPU box (test infrastructure, like the rest of tests/):
    python tests/fuzz_gpu_means.py [first_seed] [n_seeds]
Every seed draws a shape (1 .. 6000 rows, 1 .. 70 000 columns: one tile of one line up to more tiles than CUs, and now
and then more than 65 535 columns = the round-4 kernels), a density (1e-4 .. 1), a dtype, a storage (CSR, C-ordered dense,
column-major dense), row categories and a number of row pieces, and compares ``icv_colchain`` / ``icv_colsum_pairwise``
with numpy / scipy through the oracle: array_equal.  C-ordered dense float32 cases also go through the chain by integer
blocks (``ChainBlocks``: the row pieces as ranks, start estimates of random quality, scans in column ranges).  Prints the
failing seeds; exit status 1 if any."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, _p)


def case(seed):
    import numpy as np
    import scipy.sparse as sp

    rs = np.random.RandomState(seed)
    n = int(rs.choice([1, 2, 63, 64, 65, 127, 129, 500, 1000, 3000, 6000]) if rs.rand() < 0.5 else rs.randint(1, 6001))
    g = int(rs.choice([1, 5, 31, 32, 33, 96, 128, 1000, 8193, 20000, 40001, 70001], p=[.05, .05, .05, .05, .05, .1, .1, .15, .1, .15, .1, .05]))
    if n * g > 6e7:
        n = max(1, int(6e7 // g))
    dens = float(10 ** rs.uniform(-4, 0)) if rs.rand() < 0.8 else 1.0
    dtype = [np.float32, np.float64][rs.randint(2)]
    kind = rs.choice(["csr", "csr", "csr", "dense", "densef"])
    if seed >= 100000:  # seeds from 100 000 on: C-ordered dense float32 only = every case also runs the chain by blocks
        dtype, kind = np.float32, "dense"
    if kind != "csr" and g == 1:
        g = 2  # (numpy sees a single-column dense matrix as a 1-D contiguous reduction: not reproduced)
    X = rs.gamma(0.3, 1.0, (n, g)).astype(dtype)
    X[rs.rand(n, g) >= dens] = 0
    if rs.rand() < 0.2 and n > 3:
        X[rs.randint(n)] = rs.gamma(0.3, 1.0, g) + 0.1  # a full row
    if rs.rand() < 0.2 and g > 3:
        X[:, rs.randint(g)] = rs.gamma(0.3, 1.0, n) + 0.1  # a full column: 64 LDS rows a round
    if seed >= 100000:  # what the block records must hand back to the sequential replay: ties, signs, non-finite entries
        u = rs.rand()
        if u < 0.25:
            X = np.floor(X * 4).astype(dtype)  # small integers: x / ulp(s) hits one half exactly again and again
        elif u < 0.4:
            X = (X * (rs.rand(n, g) < 0.97) - X * (rs.rand(n, g) < 0.03)).astype(dtype)  # a few negative entries
        elif u < 0.5 and n > 2:
            X[rs.randint(n), rs.randint(g)] = [np.nan, np.inf, -np.inf][rs.randint(3)]
        elif u < 0.6:
            X = (X * dtype(10.0) ** rs.randint(-30, 30)).astype(dtype)  # far from 1: subnormal starts / large binades
    labels = np.array(["a", "b", "c"])[rs.randint(0, 3, n)]
    cats = None if rs.rand() < 0.5 or kind == "densef" else [["a"], ["b", "a"], ["c", "a", "b"]][rs.randint(3)]
    if (cats is not None and not all((labels == c).any() for c in cats)) or seed >= 100000:
        cats = None
    pieces = int(rs.choice([1, 1, 2, 5]))
    Xin = sp.csr_matrix(X) if kind == "csr" else (np.asfortranarray(X) if kind == "densef" else X)
    return Xin, kind, labels, cats, pieces, dict(n=n, g=g, dens=round(dens, 5), dtype=dtype.__name__, kind=kind, cats=cats, pieces=pieces)


def run(seed):
    import numpy as np
    import scipy.sparse as sp

    from infercnvpy_amd import _engine
    from oracle import infercnv_oracle as O

    Xin, kind, labels, cats, pieces, desc = case(seed)
    exp = np.asarray(O.reference_profile(Xin, None if cats is None else labels, cats, None, Xin.shape[1]))
    if kind == "densef":
        if Xin.shape[0] < 2 or Xin.shape[1] < 2:
            return desc, True  # (not a column-major matrix)
        got = _engine.fortran_column_means(Xin, max_bytes=int(4e6))[None, :]
    else:
        n = Xin.shape[0]
        dm = _engine.to_device_matrix(Xin)
        groups = [None] if cats is None else [labels == c for c in cats]
        bounds = np.linspace(0, n, pieces + 1).astype(int)
        out = []
        for sel in groups:
            count = n if sel is None else int(sel.sum())
            acc = None
            for r0, r1 in zip(bounds[:-1], bounds[1:]):
                if r1 == r0:
                    continue
                rows = None if sel is None else np.nonzero(sel[r0:r1])[0]
                acc = _engine.column_chain(dm, acc, rows, count, int(r0), int(r1))
            out.append(_engine.chain_mean(acc, count, sp.issparse(Xin)).cpu().numpy())
        got = np.vstack(out)
        if kind == "dense" and Xin.dtype == np.float32 and cats is None:
            # the same chain BY BLOCKS (icv_colchain_blocks_*: what row-sharded ranks run concurrently), the row pieces as
            # ranks; the start estimates of a random quality -- exact float64 totals, a few per cent off, off by orders of
            # magnitude, none: a wrong estimate may only cost replays, never a bit
            import torch

            rs = np.random.RandomState(seed + 7)
            acc = torch.zeros(Xin.shape[1], dtype=torch.float32, device="cuda")
            est = np.zeros(Xin.shape[1], dtype=np.float64)
            for r0, r1 in zip(bounds[:-1], bounds[1:]):
                if r1 == r0:
                    continue
                cb = _engine.ChainBlocks(dm, int(r0), int(r1))
                tot = cb.sums().cpu().numpy()
                with np.errstate(invalid="ignore", over="ignore"):
                    if not np.allclose(tot, Xin[r0:r1].sum(axis=0, dtype=np.float64), rtol=1e-11, atol=0, equal_nan=True):
                        desc["blocks"] = "float64 totals differ"
                        return desc, False
                q = rs.randint(4)
                e = [est, est * rs.uniform(0.9, 1.1, est.shape), est * 10.0 ** rs.uniform(-3, 3, est.shape), None][q]
                cb.records(None if e is None else torch.from_numpy(np.ascontiguousarray(e)).cuda())
                if rs.rand() < 0.5 and Xin.shape[1] > 64:  # in column ranges, as the pipelined hand-over scans
                    cut = int(rs.randint(1, Xin.shape[1] // 32 + 1)) * 32
                    cut = min(cut, Xin.shape[1])
                    cb.scan(acc, cols=(0, cut))
                    if cut < Xin.shape[1]:
                        cb.scan(acc, cols=(cut, Xin.shape[1]))
                else:
                    cb.scan(acc)
                est = est + tot
            got_b = _engine.chain_mean(acc, n, False).cpu().numpy()[None, :]
            desc["blocks"] = True
            if not np.array_equal(got_b, exp, equal_nan=True):
                desc["blocks"] = "MISMATCH"
                return desc, False
    return desc, got.dtype == exp.dtype and np.array_equal(got, exp, equal_nan=True)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    t0 = time.time()
    bad = []
    for seed in range(first, first + count):
        try:
            desc, ok = run(seed)
        except Exception as e:  # noqa: BLE001
            desc, ok = {"error": repr(e)}, False
        if not ok:
            bad.append((seed, desc))
            print("FAIL", seed, desc, flush=True)
    print(f"fuzz means: seeds {first}..{first + count - 1}: {count - len(bad)} passed, {len(bad)} failed {[b[0] for b in bad]} ({time.time() - t0:.0f} s)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

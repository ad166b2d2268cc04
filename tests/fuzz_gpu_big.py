"""Random cases at the BENCHMARK geometry (20 000 genes on chr1..22, random var order) against the oracle, with the
inputs the register-prefetch kernels treat specially (GPU box; test infrastructure like the rest of tests/: it checks the product against oracle/):
    python tests/fuzz_gpu_big.py [first_seed] [n_seeds]
Per case: dense float32 (k_smooth_x16 / k_smooth_ws) or CSR float32 (k_smooth_se); window 100 or 250 at step 10 (now and
then another block-form pair); 150-700 cells in chunks of 64 / 100 / 5000; lfc_clip 0.5 / 3 / 10; one or two
reference categories, a given reference or none; and rows that leave the fast path: rows equal to the reference (every
window ties: handed back to the generic kernel), all-zero rows, rows with a NaN, CSR rows with more than 2 048 stored
entries, empty CSR rows, rows of huge values (all clipped).  Compared: chr_pos, zero pattern (rounding-noise ties
excepted, see tests/test_gpu_parity.py), values to 1e-6, NaN rows."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, _p)


def one_case(seed):
    import numpy as np
    import pandas as pd
    import scipy.sparse as sp

    import cases
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    rng = np.random.RandomState(50_000 + seed)
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K, seed_start=seed, seed_perm=seed + 1)
    G = len(v["names"])
    n = int(rng.randint(150, 700))
    fmt = ["dense", "csr"][seed % 2]
    density = float(rng.choice([0.002, 0.02, 0.07, 0.2])) if fmt == "csr" else 0.19
    X = rng.gamma(0.3, 1.0, size=(n, G)).astype(np.float32)
    X[rng.rand(n, G) > density] = 0
    labels = rng.choice(["a", "b", "c"], size=n)
    labels[:3] = ["a", "b", "c"]
    win, stp = [(100, 10), (250, 10), (100, 10), (250, 10), (120, 4), (50, 10)][rng.randint(0, 6)]
    kw = dict(window_size=win, step=stp, lfc_clip=float(rng.choice([0.5, 3.0, 10.0])),
              chunksize=int(rng.choice([64, 100, 5000])), dynamic_threshold=[None, 0.5, 1.5][rng.randint(0, 3)],
              exclude_chromosomes=[("chrX", "chrY"), None, ("chr7",)][rng.randint(0, 3)])
    ref_kind = ["none", "array", "cat1", "cat2"][rng.randint(0, 4)]
    api = dict(kw)
    if ref_kind == "array":
        ref = (X.mean(axis=0) + rng.normal(0, 0.05, G)).astype(np.float32)
        api["reference"] = ref
    # special rows (after the reference is fixed for "array"; the computed references include them)
    special = {}
    rows = rng.choice(n, size=8, replace=False)
    if ref_kind == "array":
        X[rows[0]] = ref
        special["equal_to_reference"] = int(rows[0])
    X[rows[1]] = 0
    special["all_zero"] = int(rows[1])
    if rng.rand() < 0.5:
        X[rows[2], rng.randint(0, G)] = np.nan
        special["nan"] = int(rows[2])
    X[rows[3]] = rng.gamma(2.0, 1.0, size=G).astype(np.float32) + 0.01  # > 2048 stored entries
    special["dense_row"] = int(rows[3])
    X[rows[4]] = 1e6
    special["huge"] = int(rows[4])
    X[rows[5], : G // 2] = 0
    Xin = sp.csr_matrix(X) if fmt == "csr" else X
    if ref_kind == "none":
        ref = np.asarray(O.reference_profile(Xin, None, None, None, G))  # the reference's own mean of the matrix as stored
    elif ref_kind in ("cat1", "cat2"):
        cats = ["a"] if ref_kind == "cat1" else ["b", "c"]
        api.update(reference_key="group", reference_cat=cats if len(cats) > 1 else cats[0])
        ref = np.asarray(O.reference_profile(Xin, labels, cats, None, G))
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    ad = SimpleAnnData(Xin, obs=pd.DataFrame({"group": labels}), var=var)
    tm = {}
    chr_pos, res, _ = cnv.tl.infercnv(ad, inplace=False, _timings=tm, **api)
    e_pos, e_res, _, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref, **kw)
    desc = f"fmt={fmt} n={n} density={density} ref={ref_kind} kernel={tm.get('kernel')} kw={kw} special={special}"
    assert {k: int(p) for k, p in chr_pos.items()} == {k: int(p) for k, p in e_pos.items()}, desc
    got, exp = res.toarray(), e_res.toarray()
    assert got.shape == exp.shape, desc
    gn, en = np.isnan(got), np.isnan(exp)
    assert np.array_equal(gn, en), desc + f" NaN rows got {np.flatnonzero(gn.any(1))[:5]} exp {np.flatnonzero(en.any(1))[:5]}"
    got, exp = np.nan_to_num(got), np.nan_to_num(exp)
    differ = (got == 0) != (exp == 0)
    tiny = np.maximum(np.abs(got), np.abs(exp))[differ] < 1e-13
    assert np.all(tiny), desc + f" zero pattern: {int(differ.sum())} entries, rows {np.unique(np.nonzero(differ)[0])[:8]}"
    err = np.abs(got - exp).max()
    assert err <= 1e-6, desc + f" max abs error {err} at row {np.unravel_index(np.abs(got - exp).argmax(), got.shape)}"
    return desc


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = []
    t0 = time.time()
    for seed in range(first, first + n):
        try:
            d = one_case(seed)
            if os.environ.get("FUZZ_VERBOSE"):
                print(f"seed {seed} ok {d}", flush=True)
        except Exception as e:  # noqa: BLE001 -- report and go on
            bad.append(seed)
            print(f"seed {seed}: {type(e).__name__}: {str(e)[:900]}", flush=True)
    print(f"fuzz big: seeds {first}..{first + n - 1}: {n - len(bad)} passed, {len(bad)} failed {bad} "
          f"({time.time() - t0:.0f} s)", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

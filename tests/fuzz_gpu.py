"""Random cases of tests/test_gpu_parity.py's sweep over MORE seeds than the test suite runs (GPU box; test infrastructure like the rest of tests/: it checks the product against oracle/):
    python tests/fuzz_gpu.py [first_seed] [n_seeds]
Odd seeds take the block-form float32 CSR / CSC family (k_smooth_se), even seeds the general family.  Every case goes
through cnv.tl.infercnv and is compared with the oracle (chr_pos, exact zero pattern, values to 1e-6).  Prints the
failing seeds with their parameters; exit status 1 if any."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):  # as tests/conftest.py
    sys.path.insert(0, _p)


def diagnose(T, seed, sd):
    """Where a failing case differs: kernel, positions, both values, the row's median neighbourhood."""
    import numpy as np
    import pandas as pd
    import scipy.sparse as sp

    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData
    from oracle import infercnv_oracle as O

    v, X, fmt, labels, kw, ref_kind, rng = T._sweep_case(seed, sd)
    Xin = {"dense": X, "csr": sp.csr_matrix(X), "csc": sp.csc_matrix(X)}[fmt]
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    ad = SimpleAnnData(Xin, obs=pd.DataFrame({"group": labels}), var=var)
    mean_dtype = np.float32 if X.dtype == np.float32 else np.float64
    api = dict(kw)
    if ref_kind == "array":
        ref = (X.mean(axis=0) + rng.normal(0, 0.05, X.shape[1])).astype(mean_dtype)
        api["reference"] = ref
    elif ref_kind == "none":
        ref = T._oracle_means(Xin)
    else:
        cats = ["a"] if ref_kind == "cat1" else ["b", "c"]
        api.update(reference_key="group", reference_cat=cats if len(cats) > 1 else cats[0])
        ref = T._oracle_means(Xin, labels, cats)
    tm = {}
    _, res, _ = cnv.tl.infercnv(ad, inplace=False, _timings=tm, **api)
    _, e_res, _, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref, **kw)
    got, exp = res.toarray(), e_res.toarray()
    print(f"  diag: X dtype {X.dtype} shape {X.shape} kernel {tm.get('kernel')} W {got.shape[1]}", flush=True)
    rows, cols = np.nonzero((got == 0) != (exp == 0))
    for r, c in list(zip(rows, cols))[:6]:
        same = np.sum(exp[r] == exp[r, c])
        print(f"  diag: [{r},{c}] got {got[r, c]!r} exp {exp[r, c]!r}; windows of the row equal to it in the oracle: "
              f"{same}; zeros in the row got {np.sum(got[r] == 0)} exp {np.sum(exp[r] == 0)}", flush=True)
    bad = np.abs(got - exp) > 1e-6
    if bad.any():
        r, c = np.argwhere(bad)[0]
        print(f"  diag: value mismatch at [{r},{c}] got {got[r, c]!r} exp {exp[r, c]!r} ({bad.sum()} elements)", flush=True)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    import test_gpu_parity as T

    bad = []
    for seed in range(first, first + n):
        sd = bool(seed % 2)
        try:
            T._check_sweep_case(seed, sd, extras=True)
        except Exception as e:  # noqa: BLE001 -- report and go on
            _, _, fmt, _, kw, ref_kind, _ = T._sweep_case(seed, sd)
            bad.append(seed)
            print(f"seed {seed} sd={sd} fmt={fmt} ref={ref_kind} kw={kw}: {type(e).__name__}: {str(e)[:400]}", flush=True)
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
            if os.environ.get("FUZZ_DIAG"):
                diagnose(T, seed, sd)
    print(f"fuzz: seeds {first}..{first + n - 1}: {n - len(bad)} passed, {len(bad)} failed {bad}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

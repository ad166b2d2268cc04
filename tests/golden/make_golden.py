#!/usr/bin/env python
"""Generate golden vectors by RUNNING THE REFERENCE (container-only tool).

    python tests/golden/make_golden.py

Imports the reference's hot-path module from /root/reference through
``_ref_shim`` (nothing is copied), drives its public ``infercnv`` function with
a small duck-typed AnnData stand-in and stores inputs + outputs as compressed
``.npz`` fixtures next to this script.  The fixtures are data only.

The GPU box never has /root/reference; tests read only the ``.npz`` files.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import pandas as pd
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import cases  # noqa: E402
from _ref_shim import load_reference  # noqa: E402

NA = "__NA__"


class DuckAnnData:
    """Just enough AnnData for the reference driver (tl/_infercnv.py:97-158)."""

    def __init__(self, X, obs, var, layers=None):
        self.X = X
        self.obs = obs
        self.var = var
        self.layers = layers or {}
        self.obsm = {}
        self.uns = {}

    @property
    def shape(self):
        return self.X.shape

    @property
    def var_names(self):
        return self.var.index

    def __getitem__(self, key):
        rows, cols = key
        assert isinstance(rows, slice) and rows == slice(None)
        cols = np.asarray(cols)
        X = self.X[:, cols]
        layers = {k: v[:, cols] for k, v in self.layers.items()}
        return DuckAnnData(X, self.obs, self.var.loc[cols], layers)


def make_adata(X, var, obs_labels=None):
    chrom = [None if c is None else c for c in var["chromosome"]]
    vdf = pd.DataFrame({"chromosome": chrom, "start": var["start"], "end": var["end"]}, index=var["names"])
    obs = pd.DataFrame(index=[f"cell{i}" for i in range(X.shape[0])])
    if obs_labels is not None:
        obs["group"] = obs_labels
    return DuckAnnData(X, obs, vdf)


def run_case(ref, name, X, var, kwargs, obs_labels=None, store_input=True, fmt="dense", in_seed=None):
    if ONLY is not None and not name.startswith(ONLY):
        return
    Xin = X
    if fmt == "csr":
        Xin = sp.csr_matrix(X)
    elif fmt == "csc":
        Xin = sp.csc_matrix(X)
    elif fmt == "dense_f":  # column-major storage: numpy reduces np.mean(X, axis=0) pairwise per column (:385)
        Xin = np.asfortranarray(X)
    adata = make_adata(Xin, var, obs_labels)
    kw = dict(kwargs)
    kw["inplace"] = False
    kw.setdefault("n_jobs", 1)
    chr_pos, res, per_gene = ref.infercnv(adata, **kw)
    out = dict(
        out=np.asarray(res.toarray(), dtype=np.float64),
        chr_names=np.array(list(chr_pos.keys())),
        chr_vals=np.array([int(v) for v in chr_pos.values()], dtype=np.int64),
        chromosome=np.array([NA if c is None else c for c in var["chromosome"]]),
        start=np.asarray(var["start"]),
        fmt=np.array(fmt),
        in_dtype=np.array(str(X.dtype)),
        in_shape=np.array(X.shape),
        in_checksum=np.array(cases.checksum(X)),
    )
    if store_input:
        out["X"] = X
    elif in_seed is not None:
        out["in_seed"] = np.array(in_seed)
    if obs_labels is not None:
        out["obs"] = np.array(obs_labels)
    if per_gene is not None:
        out["per_gene"] = per_gene
    jk = {}
    for k, v in kwargs.items():
        if isinstance(v, np.ndarray):
            out["kw_" + k] = v
        else:
            jk[k] = v
    out["kwargs"] = np.array(json.dumps(jk))
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    nnz = np.count_nonzero(out["out"])
    print(f"{name:28s} X{X.shape} {X.dtype} {fmt:5s} -> out{out['out'].shape} nnz={nnz} "
          f"chr_pos={dict(zip(out['chr_names'], out['chr_vals']))}  [{os.path.getsize(path) / 1024:.0f} KiB]")


class DuckRows:
    """AnnData stand-in for the score functions: boolean row selection only (tl/_scores.py:131, :201)."""

    def __init__(self, X, obs, obsm):
        self.X, self.obs, self.obsm = X, obs, obsm

    @property
    def shape(self):
        return self.X.shape

    def __getitem__(self, key):
        rows, _ = key
        m = np.asarray(rows)
        return DuckRows(self.X[m], self.obs[m], {k: v[m] for k, v in self.obsm.items()})


def ith_cases(scores):
    """ithgex / ithcna through the reference's own functions on seeded data."""
    rs = np.random.RandomState(11)
    n, k = 400, 333
    labels = np.array(["g0"] * 150 + ["g1"] * 249 + ["solo"])[rs.permutation(n)]
    base = rs.standard_normal((3, k))
    X = (rs.standard_normal((n, k)) + 0.7 * base[rs.randint(0, 3, n)]).astype(np.float32)
    X[labels == "g1"] *= rs.gamma(2.0, 1.0, (int((labels == "g1").sum()), 1)).astype(np.float32)
    cnv = sp.csr_matrix(np.where(np.abs(X) > 0.8, X, 0).astype(np.float64))
    obs = pd.DataFrame({"group": labels})
    ad = DuckRows(X, obs, {"X_cnv": cnv})
    gex = scores.ithgex(ad, "group", inplace=False)
    cna = scores.ithcna(ad, "group", inplace=False)
    keys = sorted(gex)
    assert keys == sorted(cna) == ["g0", "g1"]
    np.savez_compressed(os.path.join(HERE, "ith_scores.npz"), X=X, labels=labels, keys=np.array(keys),
                        gex=np.array([gex[g] for g in keys]), cna=np.array([cna[g] for g in keys]))
    print("ith_scores:", {g: (gex[g], cna[g]) for g in keys})


ONLY = None  # --only PREFIX[,PREFIX...]: write just the cases whose name starts with one of the prefixes


def main():
    global ONLY
    ref, scores = load_reference()
    if "--only-ith" in sys.argv:
        return ith_cases(scores)
    if "--only" in sys.argv:
        ONLY = tuple(sys.argv[sys.argv.index("--only") + 1].split(","))
    else:
        ith_cases(scores)

    extra = (("chrX", 30), ("chrY", 6), ("chrM", 8), ("GL000218.1", 9), (None, 5))
    var_m = cases.synthetic_var([230, 110, 101, 100, 99, 57, 140],
                                names=["chr1", "chr2", "chr3", "chr4", "chr5", "chr6", "chr10"], extra=extra)
    G = len(var_m["names"])
    C = 48
    Xf = cases.synthetic_expr(C, G, seed=2, dtype=np.float32)
    ref32 = Xf.mean(axis=0, dtype=np.float64).astype(np.float32)
    labels = np.array((["normalA"] * 10 + ["normalB"] * 9 + ["tumor"] * 29))
    labels = labels[np.random.RandomState(3).permutation(C)]

    run_case(ref, "m_dense_f32_r1", Xf, var_m, dict(reference=ref32))
    run_case(ref, "m_csr_f32_r1", Xf, var_m, dict(reference=ref32), fmt="csr")
    run_case(ref, "m_csc_f32_r1", Xf, var_m, dict(reference=ref32), fmt="csc")
    run_case(ref, "m_dense_f32_allmean", Xf, var_m, dict())
    run_case(ref, "m_csr_f32_allmean", Xf, var_m, dict(), fmt="csr")
    run_case(ref, "m_dense_f32_r2", Xf, var_m, dict(reference_key="group", reference_cat=["normalA", "normalB"]),
             obs_labels=labels)
    run_case(ref, "m_csr_f32_r2", Xf, var_m, dict(reference_key="group", reference_cat=["normalA", "normalB"]),
             obs_labels=labels, fmt="csr")
    run_case(ref, "m_dense_f32_r1cat", Xf, var_m, dict(reference_key="group", reference_cat="normalA"),
             obs_labels=labels)
    ref2 = np.vstack([Xf[labels == "normalA"].mean(axis=0, dtype=np.float64),
                      Xf[labels == "normalB"].mean(axis=0, dtype=np.float64)]).astype(np.float32)
    run_case(ref, "m_dense_f32_r2given", Xf, var_m, dict(reference=ref2))

    Xd = cases.synthetic_expr(C, G, seed=4, dtype=np.float64)
    run_case(ref, "m_dense_f64_r1", Xd, var_m, dict(reference=Xd.mean(axis=0)))
    run_case(ref, "m_csr_f64_r1", Xd, var_m, dict(reference=Xd.mean(axis=0)), fmt="csr")
    # float32 matrix with a float64 reference -> numpy promotes the subtraction to float64
    run_case(ref, "m_dense_f32_ref64", Xf, var_m, dict(reference=Xf.mean(axis=0, dtype=np.float64)))

    # column-major dense input (np.asfortranarray / a transposed genes x cells array): numpy's all-cell mean is pairwise
    # per column, over 8 192-element pieces of the iterator's buffer (the 9 000-cell case crosses a piece boundary);
    # per-category means go through X[rows, :], which numpy returns C-ordered
    run_case(ref, "m_densef_f32_allmean", Xf, var_m, dict(), fmt="dense_f")
    run_case(ref, "m_densef_f64_allmean", Xd_f := cases.synthetic_expr(C, G, seed=4, dtype=np.float64), var_m, dict(),
             fmt="dense_f")
    run_case(ref, "m_densef_f32_r2", Xf, var_m, dict(reference_key="group", reference_cat=["normalA", "normalB"]),
             obs_labels=labels, fmt="dense_f")
    var_f = cases.synthetic_var([230, 110, 101], names=["chr1", "chr2", "chr3"], seed_start=30, seed_perm=31)
    Xbig = cases.synthetic_expr(9000, len(var_f["names"]), seed=14)
    run_case(ref, "m_densef_f32_9000", Xbig, var_f, dict(chunksize=3000), fmt="dense_f", store_input=False, in_seed=14)

    Xi = cases.synthetic_counts(C, G, seed=5)
    run_case(ref, "m_densef_i64_allmean", Xi, var_m, dict(), fmt="dense_f")
    run_case(ref, "m_dense_i64_allmean", Xi, var_m, dict())
    run_case(ref, "m_dense_i64_r2", Xi, var_m, dict(reference_key="group", reference_cat=["normalA", "normalB"]),
             obs_labels=labels)
    run_case(ref, "m_csr_i64_r2", Xi, var_m, dict(reference_key="group", reference_cat=["normalA", "normalB"]),
             obs_labels=labels, fmt="csr")

    run_case(ref, "m_w101_s7", Xf, var_m, dict(reference=ref32, window_size=101, step=7))
    run_case(ref, "m_w3_s1", Xf, var_m, dict(reference=ref32, window_size=3, step=1))
    run_case(ref, "m_w100_s1", Xf, var_m, dict(reference=ref32, window_size=100, step=1))
    run_case(ref, "m_w50_s10", Xf, var_m, dict(reference=ref32, window_size=50, step=10))
    run_case(ref, "m_w20_s4", Xf, var_m, dict(reference=ref32, window_size=20, step=4))
    run_case(ref, "m_w100_s3", Xf, var_m, dict(reference=ref32, window_size=100, step=3))
    run_case(ref, "m_nothr", Xf, var_m, dict(reference=ref32, dynamic_threshold=None))
    run_case(ref, "m_thr05", Xf, var_m, dict(reference=ref32, dynamic_threshold=0.5))
    run_case(ref, "m_chunks20", Xf, var_m, dict(reference=ref32, chunksize=20))
    run_case(ref, "m_csr_chunks20", Xf, var_m, dict(reference=ref32, chunksize=20), fmt="csr")
    run_case(ref, "m_clip05", Xf, var_m, dict(reference=ref32, lfc_clip=0.5))
    run_case(ref, "m_clip01", Xf, var_m, dict(reference=ref32, lfc_clip=0.1))
    run_case(ref, "m_exclude_none", Xf, var_m, dict(reference=ref32, exclude_chromosomes=None))
    run_case(ref, "m_exclude_chr2", Xf, var_m, dict(reference=ref32, exclude_chromosomes=["chr2", "chrX"]))

    # window 250 needs longer chromosomes
    var_w = cases.synthetic_var([600, 260, 251, 250, 249], seed_start=10, seed_perm=11)
    Xw = cases.synthetic_expr(32, len(var_w["names"]), seed=6)
    refw = Xw.mean(axis=0, dtype=np.float64).astype(np.float32)
    run_case(ref, "w250_s10", Xw, var_w, dict(reference=refw, window_size=250, step=10))
    run_case(ref, "w250_s10_csr", Xw, var_w, dict(reference=refw, window_size=250, step=10), fmt="csr")
    run_case(ref, "w250_s5", Xw, var_w, dict(reference=refw, window_size=250, step=5))

    # CSR input in block form = the stored-entries kernel (k_smooth_se): bounded references (:424-432), other clip
    # values (its fixed-point scale), other block sizes; columns without a chromosome / on chrX, a chromosome with
    # fewer genes than the window (flat window, :227-236), one with exactly `window` genes
    var_s = cases.synthetic_var([600, 260, 251, 250, 249, 90], seed_start=12, seed_perm=13,
                                extra=(("chrX", 31), ("chrM", 3), (None, 4)))
    Xs = cases.synthetic_expr(40, len(var_s["names"]), seed=9)
    refs = Xs.mean(axis=0, dtype=np.float64).astype(np.float32)
    labs = np.array(["nA"] * 9 + ["nB"] * 8 + ["t"] * 23)[np.random.RandomState(5).permutation(40)]
    w250 = dict(window_size=250, step=10)
    run_case(ref, "w250_s10_csr_r2", Xs, var_s, dict(reference_key="group", reference_cat=["nA", "nB"], **w250),
             obs_labels=labs, fmt="csr")
    refs2 = np.vstack([Xs[labs == "nA"].mean(axis=0, dtype=np.float64),
                       Xs[labs == "nB"].mean(axis=0, dtype=np.float64)]).astype(np.float32)
    run_case(ref, "w250_s10_csr_r2given", Xs, var_s, dict(reference=refs2, **w250), fmt="csr")
    run_case(ref, "w250_s10_csr_clip05", Xs, var_s, dict(reference=refs, lfc_clip=0.5, **w250), fmt="csr")
    run_case(ref, "w250_s10_csr_clip10", Xs, var_s, dict(reference=refs, lfc_clip=10, **w250), fmt="csr")
    run_case(ref, "w120_s4_csr", Xs, var_s, dict(reference=refs, window_size=120, step=4, chunksize=16), fmt="csr")
    run_case(ref, "w100_s2_csr", Xs, var_s, dict(reference=refs, window_size=100, step=2), fmt="csr")
    run_case(ref, "w100_s10_csr_x", Xs, var_s, dict(reference=refs, window_size=100, step=10), fmt="csr")
    run_case(ref, "w100_s10_csc_r2", Xs, var_s, dict(reference_key="group", reference_cat=["nB", "nA"],
                                                      window_size=100, step=10, dynamic_threshold=1.0),
             obs_labels=labs, fmt="csc")

    # GTF-ordered input (identity gather) and a dense, no-zeros matrix
    var_o = cases.synthetic_var([230, 110, 101, 100, 99, 57, 140],
                                names=["chr1", "chr2", "chr3", "chr4", "chr5", "chr6", "chr10"], permute=False)
    Xo = np.random.RandomState(8).normal(1.0, 0.7, size=(24, len(var_o["names"]))).astype(np.float32)
    run_case(ref, "ordered_normal", Xo, var_o, dict(reference=Xo[:8].mean(axis=0)))

    # per-gene values (tiny: the reference's python dict loops are O(C*W*n))
    var_g = cases.synthetic_var([25, 12, 10, 7], seed_start=20, seed_perm=21, extra=(("chrX", 3), (None, 2)))
    Xg = cases.synthetic_expr(6, len(var_g["names"]), seed=7)
    refg = Xg.mean(axis=0, dtype=np.float64).astype(np.float32)
    run_case(ref, "genevals_w10_s3", Xg, var_g,
             dict(reference=refg, window_size=10, step=3, calculate_gene_values=True, chunksize=4))
    run_case(ref, "genevals_w10_s1", Xg, var_g,
             dict(reference=refg, window_size=10, step=1, calculate_gene_values=True))

    # full benchmark geometry, input stored by checksum only
    var_b = cases.synthetic_var(cases.GENES_PER_CHROM_20K, seed_start=0, seed_perm=1)
    Xb = cases.synthetic_expr(96, 20000, seed=2)
    refb = Xb.mean(axis=0, dtype=np.float64).astype(np.float32)
    run_case(ref, "big20k_w100_s10", Xb, var_b, dict(reference=refb, chunksize=64), store_input=False)
    run_case(ref, "big20k_w250_s10_csr", Xb[:40], var_b, dict(reference=refb, window_size=250, chunksize=5000),
             store_input=False, fmt="csr")

    # the reference's own seeded 4 x 10 fixture, through the driver (chunksize=2)
    x, chrom, start, names = cases.adata_full_mock_arrays()
    var4 = dict(chromosome=chrom, start=start, end=start + 99, names=names)
    run_case(ref, "mock4x10_chunks2", x, var4,
             dict(chunksize=2, lfc_clip=1, window_size=3, step=1, dynamic_threshold=1, calculate_gene_values=True),
             fmt="csr")

    # reference means in isolation (dense / CSR, float32): _get_reference
    for fmt in ("dense", "csr") if ONLY is None else ():
        Xin = sp.csr_matrix(Xf) if fmt == "csr" else Xf
        ad = make_adata(Xin, var_m, labels)
        r_all = np.asarray(ref._get_reference(ad, None, None, None, None))
        r_cat = np.asarray(ref._get_reference(ad, "group", ["normalA", "normalB"], None, None))
        np.savez_compressed(os.path.join(HERE, f"refmean_{fmt}.npz"), X=Xf, obs=labels, r_all=r_all, r_cat=r_cat)
        print(f"refmean_{fmt}: {r_all.dtype} {r_all.shape} {r_cat.shape}")


if __name__ == "__main__":
    main()

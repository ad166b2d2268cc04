"""Deterministic input builders shared by ``make_golden.py`` (container-only,
runs the reference) and the test-suite (runs everywhere).

Inputs of the small cases are ALSO stored inside the ``.npz`` fixtures, so the
tests never depend on RNG stream stability; the builders here exist so that the
generator script and the tests agree on what each case means, and for the
larger seeded cases whose inputs are too big to commit (those fixtures carry an
input checksum instead).
"""
from __future__ import annotations

import numpy as np

# 20 000 genes on chr1..chr22
GENES_PER_CHROM_20K = [2340, 1330, 1110, 780, 910, 1080, 950, 710, 810, 760, 1330, 1060, 340, 680,
                       630, 900, 1210, 290, 1500, 570, 250, 460]


def synthetic_var(genes_per_chrom, names=None, seed_start=0, seed_perm=1, permute=True, extra=()):
    """Gene annotation: unique integer starts per chromosome, optional random var order.

    ``extra``: iterable of (chromosome_name_or_None, n_genes) appended before permutation
    (e.g. ("chrX", 30), ("chrM", 5), ("GL000", 7), (None, 4)).
    Returns dict(chromosome=object array, start=int64 array, end=int64 array, names=list[str]).
    """
    if names is None:
        names = [f"chr{i + 1}" for i in range(len(genes_per_chrom))]
    rng = np.random.RandomState(seed_start)
    chrom, start = [], []
    for name, g in list(zip(names, genes_per_chrom)) + list(extra):
        pos = rng.permutation(50 * max(g, 1))[:g] * 100 + 7
        chrom += [name] * g
        start += list(pos)
    chrom = np.array(chrom, dtype=object)
    start = np.array(start, dtype=np.int64)
    if permute:
        p = np.random.RandomState(seed_perm).permutation(len(chrom))
        chrom, start = chrom[p], start[p]
    gene_names = [f"g{i}" for i in range(len(chrom))]
    return dict(chromosome=chrom, start=start, end=start + 1000, names=gene_names)


def position_ordered(v, chrom_order=None):
    """The same annotation with ``var`` in GENOME order (what a GTF-derived ``adata.var`` looks like): chromosomes one
    after the other (``chrom_order``: list of names, default first appearance in natural order; genes without a
    chromosome last), positions ascending inside each.  Returns (v_sorted, perm) with ``v_sorted[k] = v[k][perm]``."""
    chrom, start = np.asarray(v["chromosome"], dtype=object), np.asarray(v["start"])
    names = [c for c in dict.fromkeys(chrom.tolist()) if c is not None]
    if chrom_order is None:
        import re

        chrom_order = sorted(names, key=lambda k: [int(x) if x.isdigit() else x.lower() for x in re.split("([0-9]+)", k)])
    rank = {c: i for i, c in enumerate(chrom_order)}
    key = np.array([rank.get(c, len(rank)) for c in chrom.tolist()])
    perm = np.lexsort((start, key))
    out = {k: (np.asarray(x, dtype=object)[perm] if k == "chromosome" else
               ([x[i] for i in perm] if isinstance(x, list) else np.asarray(x)[perm])) for k, x in v.items()}
    return out, perm


def synthetic_expr(n_cells, n_genes, seed=2, dtype=np.float32, density_cut=0.5):
    """log1p-like expression: gamma(0.3, 1), entries < cut set to 0 (~19 % nnz).  SURVEY §8(d)."""
    rng = np.random.RandomState(seed)
    x = rng.gamma(0.3, 1.0, size=(n_cells, n_genes))
    x[x < density_cut] = 0
    return x.astype(dtype)


def synthetic_counts(n_cells, n_genes, seed=5):
    """Small integer counts (for the integer-dtype semantics cases)."""
    rng = np.random.RandomState(seed)
    return rng.poisson(0.8, size=(n_cells, n_genes)).astype(np.int64)


def adata_full_mock_arrays():
    """The reference's 4 x 10 seeded fixture (reference tests/conftest.py:61-75)."""
    np.random.seed(0)
    x = np.random.randint(low=0, high=50, size=(4, 10))
    chrom = np.array(["chr1"] * 5 + ["chr2"] * 5, dtype=object)
    start = np.array([100, 200, 300, 400, 500, 0, 100, 200, 300, 400])
    names = [f"gene{i}" for i in range(1, 11)]
    return x, chrom, start, names


def checksum(a) -> str:
    import hashlib

    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.view(np.uint8)).hexdigest()[:16]

"""Loader for the committed golden fixtures (tests/golden/*.npz)."""
from __future__ import annotations

import glob
import json
import os

import numpy as np
import scipy.sparse as sp

import cases

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NA = "__NA__"


def case_names(prefixes=None, exclude=("refmean_", "ith_")):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    names = [n for n in names if not any(n.startswith(e) for e in exclude)]
    if prefixes:
        names = [n for n in names if any(n.startswith(p) for p in prefixes)]
    return names


class GoldenCase:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.out = z["out"]
        self.chr_pos = {str(k): int(v) for k, v in zip(z["chr_names"], z["chr_vals"])}
        self.chromosome = np.array([None if c == NA else str(c) for c in z["chromosome"]], dtype=object)
        self.start = z["start"]
        self.fmt = str(z["fmt"])
        self.kwargs = json.loads(str(z["kwargs"]))
        for k in z.files:
            if k.startswith("kw_"):
                self.kwargs[k[3:]] = z[k]
        self.obs = z["obs"] if "obs" in z.files else None
        self.per_gene = z["per_gene"] if "per_gene" in z.files else None
        self.in_dtype = str(z["in_dtype"])
        if "X" in z.files:
            self.X_dense = z["X"]
        elif "in_seed" in z.files:  # stored by seed (tests/golden/make_golden.py: in_seed)
            shape = tuple(int(v) for v in z["in_shape"])
            self.X_dense = cases.synthetic_expr(shape[0], shape[1], seed=int(z["in_seed"]))
            assert cases.checksum(self.X_dense) == str(z["in_checksum"]), "seeded input drifted"
        else:
            self.X_dense = self._rebuild(tuple(int(v) for v in z["in_shape"]))
            assert cases.checksum(self.X_dense) == str(z["in_checksum"]), "seeded input drifted"
        if self.kwargs.get("exclude_chromosomes", 0) is not None and "exclude_chromosomes" in self.kwargs:
            self.kwargs["exclude_chromosomes"] = tuple(self.kwargs["exclude_chromosomes"])

    def _rebuild(self, shape):
        # only the big20k cases are stored by seed (tests/golden/make_golden.py)
        x = cases.synthetic_expr(96, 20000, seed=2)
        return x[: shape[0]]

    @property
    def X(self):
        if self.fmt == "csr":
            return sp.csr_matrix(self.X_dense)
        if self.fmt == "csc":
            return sp.csc_matrix(self.X_dense)
        if self.fmt == "dense_f":
            return np.asfortranarray(self.X_dense)
        return self.X_dense

    def api_kwargs(self):
        """kwargs for an infercnv(adata, ...) style call (obs column name 'group')."""
        return dict(self.kwargs)

    def array_kwargs(self):
        """kwargs for oracle.infercnv(X, chrom, start, ...)."""
        kw = dict(self.kwargs)
        key = kw.pop("reference_key", None)
        kw["obs_col"] = self.obs if key is not None else None
        return kw

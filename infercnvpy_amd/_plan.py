"""Host-side planning: gene/position indexing contract + native plan handle.

Restates (not copies) the bookkeeping of the reference driver and its per-chromosome gather
(icbi-lab/infercnvpy ``src/infercnvpy/tl/_infercnv.py``):

* ``:104-108``  genes without a chromosome, or on an excluded chromosome, are masked;
* ``:327``      only chromosome names that start with ``"chr"`` and are not ``"chrM"`` get windows,
                in natural order (``chr2`` before ``chr10``, ``:164-176``);
* ``:350-351``  within a chromosome genes are ordered by ``start`` (pandas ``sort_values`` =
                numpy quicksort ``argsort``; NaN starts last);
* ``:335-337``  ``chr_pos`` = first window index of every chromosome.

The product of this module is ``col_pos`` (input column -> position in the chromosome-sorted
concatenation, -1 = masked) and ``chrom_offsets``; the native library turns them into the window
table, the padded LDS layout and the kernel variant (``csrc/icv_plan.hpp``).
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np

from . import _lib


def natural_key(name: str):
    return [int(tok) if tok.isdigit() else tok.lower() for tok in re.split(r"([0-9]+)", name)]


def _null_mask(values: np.ndarray) -> np.ndarray:
    """pandas-style isnull for an object / float / string column (None, NaN, pd.NA, NaT)."""
    if values.dtype.kind == "f":
        return np.isnan(values)
    if values.dtype.kind == "O":
        import pandas as pd

        return np.asarray(pd.isna(values), dtype=bool)
    return np.zeros(len(values), dtype=bool)


def _sorted_positions(idx: np.ndarray, start: np.ndarray) -> np.ndarray:
    s = start[idx]
    if s.dtype.kind == "f":
        nan = np.isnan(s)
        head = idx[~nan]
        return np.concatenate([head[np.argsort(s[~nan], kind="quicksort")], idx[nan]])
    return idx[np.argsort(s, kind="quicksort")]


class GenePlan:
    """Window plan for one (var annotation, window, step) combination."""

    def __init__(self, chromosome, start, *, window_size: int, step: int, exclude_chromosomes=("chrX", "chrY")):
        chrom = np.asarray(chromosome)
        if chrom.dtype.kind not in "OUS":
            chrom = chrom.astype(object)
        chrom = chrom.astype(object)
        start = np.asarray(start)
        if start.dtype.kind == "O":  # e.g. a nullable integer column with pd.NA: missing starts sort last (NaN)
            import pandas as pd

            start = pd.to_numeric(pd.Series(start), errors="coerce").to_numpy(dtype=np.float64, na_value=np.nan)
        n_all = len(chrom)
        if len(start) != n_all:
            raise ValueError("chromosome and start must have the same length")

        null = _null_mask(chrom)
        self.n_without_position = int(null.sum())
        masked = null.copy()
        if exclude_chromosomes is not None:
            excl = set(exclude_chromosomes)
            masked |= np.fromiter((c in excl for c in chrom), dtype=bool, count=n_all)
        self.var_mask = masked  # True = column not used at all (reference var_mask)

        kept = np.flatnonzero(~masked)
        names = []
        seen = set()
        for c in chrom[kept]:
            if c not in seen:
                seen.add(c)
                names.append(c)
        names = [c for c in names if isinstance(c, str) and c.startswith("chr") and c != "chrM"]
        self.chromosomes = sorted(names, key=natural_key)

        col_pos = np.full(n_all, -1, dtype=np.int32)
        offsets = [0]
        order = []
        kept_chrom = chrom[kept]
        for name in self.chromosomes:
            idx = kept[np.flatnonzero(kept_chrom == name)]
            idx = _sorted_positions(idx, start)
            order.append(idx)
            offsets.append(offsets[-1] + len(idx))
        self.order = np.concatenate(order).astype(np.int64) if order else np.zeros(0, dtype=np.int64)
        col_pos[self.order] = np.arange(len(self.order), dtype=np.int32)
        self.col_pos = col_pos
        self.chrom_offsets = np.asarray(offsets, dtype=np.int32)
        self.n_cols_all = n_all
        self.window_size = int(window_size)
        self.step = int(step)

        if not self.chromosomes:
            # the reference fails here as well (zip(*[]) in _running_mean_by_chromosome, :333)
            raise ValueError("No chromosome with a name starting with 'chr' (other than chrM) has genes.")

        lib = _lib.load()
        handle = C.c_void_p()
        _lib.check(lib.icv_plan_create(
            n_all, self.col_pos.ctypes.data, len(self.chromosomes), self.chrom_offsets.ctypes.data,
            self.window_size, self.step, C.byref(handle)))
        self._handle = handle
        info = _lib.PlanInfo()
        _lib.check(lib.icv_plan_get_info(handle, C.byref(info)))
        self.info = info
        self.n_windows = int(info.n_windows)
        pos = np.zeros(len(self.chromosomes), dtype=np.int32)
        _lib.check(lib.icv_plan_chr_pos(handle, pos.ctypes.data))
        # plain {str: np.int64}, chromosome order = natural order (reference :335-337)
        self.chr_pos = {name: np.int64(p) for name, p in zip(self.chromosomes, pos)}

    @property
    def handle(self):
        return self._handle

    def window_table(self):
        """(start, length) of every window in chromosome-sorted gene coordinates."""
        st = np.zeros(self.n_windows, dtype=np.int32)
        ln = np.zeros(self.n_windows, dtype=np.int32)
        _lib.check(_lib.load().icv_plan_window_table(self._handle, st.ctypes.data, ln.ctypes.data))
        return st, ln

    def se_tables(self):
        """Host tables of the CSR stored-entries kernel (``icv_plan_se_tables``), or None if the geometry does not
        admit it: dict(col_block, col_offset, block_gene0, w0, w1)."""
        lib = _lib.load()
        ok = C.c_int32(0)
        _lib.check(lib.icv_plan_se_tables(self._handle, C.byref(ok), None, None, None, None, None))
        if not ok.value:
            return None
        cb = np.zeros(self.n_cols_all, dtype=np.int32)
        co = np.zeros(self.n_cols_all, dtype=np.int32)
        g0 = np.zeros(int(self.info.n_blocks), dtype=np.int32)
        w0 = np.zeros(self.n_windows, dtype=np.uint32)
        w1 = np.zeros(self.n_windows, dtype=np.uint32)
        _lib.check(lib.icv_plan_se_tables(self._handle, C.byref(ok), cb.ctypes.data, co.ctypes.data, g0.ctypes.data,
                                          w0.ctypes.data, w1.ctypes.data))
        return dict(col_block=cb, col_offset=co, block_gene0=g0, w0=w0, w1=w1)

    def gene_runs(self):
        """Host tables of ``calculate_gene_values`` (``icv_plan_gene_runs``): dict(first, count, genes, col_run) -- run r
        averages the windows ``[first[r], first[r] + count[r])`` for ``genes[r]`` genes; ``col_run[c]`` is the run of
        input column c (-1: the gene has no value)."""
        lib = _lib.load()
        n = C.c_int32(0)
        _lib.check(lib.icv_plan_gene_runs(self._handle, C.byref(n), None, None, None, None))
        first, count, genes = (np.zeros(n.value, dtype=np.int32) for _ in range(3))
        col_run = np.zeros(self.n_cols_all, dtype=np.int32)
        _lib.check(lib.icv_plan_gene_runs(self._handle, C.byref(n), first.ctypes.data, count.ctypes.data,
                                          genes.ctypes.data, col_run.ctypes.data))
        return dict(first=first, count=count, genes=genes, col_run=col_run)

    def last_kernel(self) -> int:
        """``_lib.ICV_KERNEL_*`` of the smoothing kernel the last compute call on this plan launched."""
        kind = C.c_int32(0)
        _lib.check(_lib.load().icv_plan_last_kernel(self._handle, C.byref(kind)))
        return int(kind.value)

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            _lib.load().icv_plan_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

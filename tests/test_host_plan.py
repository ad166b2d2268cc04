"""CPU tests: the C-ABI library loads, exports every declared symbol, and the host-side plan
(gene order, window table, chr_pos) agrees with the oracle and with pandas' sort."""
import ctypes
import os
import re

import numpy as np
import pandas as pd
import pytest

import cases
from _golden import GoldenCase, case_names
from infercnvpy_amd import _lib
from infercnvpy_amd._plan import GenePlan
from oracle import infercnv_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "infercnv_hip.h")).read()
    declared = set(re.findall(r"\b(icv_[a-z_0-9]+)\s*\(", header)) - {"icv_status"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.icv_version() >= 100
    assert lib.icv_device_count() >= 0


def test_plan_errors_map_to_value_error():
    with pytest.raises(ValueError):
        GenePlan(np.array(["GL1", "GL1"], dtype=object), np.array([1, 2]), window_size=10, step=1)
    lib = _lib.load()
    h = ctypes.c_void_p()
    col_pos = np.array([0, 0, 1], dtype=np.int32)  # not a permutation
    off = np.array([0, 2], dtype=np.int32)
    rc = lib.icv_plan_create(3, col_pos.ctypes.data, 1, off.ctypes.data, 2, 1, ctypes.byref(h))
    assert rc == _lib.ICV_ERR_INVALID
    assert b"permutation" in lib.icv_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


@pytest.mark.parametrize("name", case_names())
def test_plan_matches_reference_chr_pos(name):
    g = GoldenCase(name)
    kw = g.kwargs
    plan = GenePlan(g.chromosome, g.start, window_size=kw.get("window_size", 100), step=kw.get("step", 10),
                    exclude_chromosomes=kw.get("exclude_chromosomes", ("chrX", "chrY")))
    assert {k: int(v) for k, v in plan.chr_pos.items()} == g.chr_pos
    assert list(plan.chr_pos.keys()) == list(g.chr_pos.keys())
    assert plan.n_windows == g.out.shape[1]
    # gene order = the oracle's (and hence the reference's) per-chromosome order
    keep = ~plan.var_mask
    ch, st = g.chromosome[keep], g.start[keep]
    kept_idx = np.flatnonzero(keep)
    expect = np.concatenate([kept_idx[O.chromosome_gene_order(ch, st, c)] for c in O.used_chromosomes(ch)])
    np.testing.assert_array_equal(plan.order, expect)
    # window table is consistent with W_c = ceil((G_c - n + 1) / step) or 1
    st_w, ln_w = plan.window_table()
    n, s = plan.window_size, plan.step
    w = 0
    for c, name_c in enumerate(plan.chromosomes):
        gc = plan.chrom_offsets[c + 1] - plan.chrom_offsets[c]
        wc = -(-(gc - n + 1) // s) if n < gc else 1
        assert plan.chr_pos[name_c] == w
        for j in range(wc):
            assert st_w[w + j] == plan.chrom_offsets[c] + (j * s if n < gc else 0)
            assert ln_w[w + j] == (n if n < gc else gc)
        w += wc
    assert w == plan.n_windows


def test_gene_order_matches_pandas_sort_values_with_ties_and_nan():
    rng = np.random.RandomState(0)
    n = 400
    chrom = rng.choice(["chr1", "chr2", "chr10", "chrX", "chrM", "KI27", None], size=n).astype(object)
    start = rng.randint(0, 40, size=n).astype(float)  # many ties
    start[rng.rand(n) < 0.05] = np.nan
    var = pd.DataFrame({"chromosome": chrom, "start": start}, index=[f"g{i}" for i in range(n)])
    plan = GenePlan(var["chromosome"].to_numpy(), var["start"].to_numpy(), window_size=10, step=2)
    assert plan.chromosomes == ["chr1", "chr2", "chr10"]
    expect = []
    for c in plan.chromosomes:
        genes = var.loc[var["chromosome"] == c].sort_values("start").index.to_numpy()
        expect.append(var.index.get_indexer(genes))
    np.testing.assert_array_equal(plan.order, np.concatenate(expect))
    assert plan.n_without_position == int(pd.isnull(var["chromosome"]).sum())


def test_plan_info_benchmark_geometry():
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    plan = GenePlan(v["chromosome"], v["start"], window_size=100, step=10)
    assert plan.n_windows == 1802
    assert plan.info.block == 10 and plan.info.n_blocks == 2000 and plan.info.padded_len == 20000
    assert plan.info.workgroups_per_cu_f32 == 2, plan.info.lds_bytes_f32
    plan250 = GenePlan(v["chromosome"], v["start"], window_size=250, step=10)
    assert plan250.n_windows == 1472 and plan250.info.block == 5
    plan1 = GenePlan(v["chromosome"], v["start"], window_size=100, step=1)
    assert plan1.n_windows == 17822 and plan1.info.block == 1


def test_plan_accepts_pandas_nullable_and_categorical_columns():
    """`adata.var` columns as pandas gives them: Categorical chromosome with missing values, nullable Int64 start
    with pd.NA (the reference's `sort_values("start")` puts missing starts last, tl/_infercnv.py:350)."""
    import pandas as pd
    from infercnvpy_amd._plan import GenePlan

    chrom = ["chr1", "chr2", None, "chr1", "chr1", "chr2", "chrX", "chr1"]
    start = [30, 5, 7, None, 10, 1, 2, 20]
    plain = GenePlan(np.array(chrom, dtype=object), np.array([np.nan if s is None else s for s in start]),
                     window_size=2, step=1)
    df = pd.DataFrame({"chromosome": pd.Categorical(chrom), "start": pd.array(start, dtype="Int64")})
    fancy = GenePlan(df["chromosome"].to_numpy(), df["start"].to_numpy(), window_size=2, step=1)
    assert fancy.chromosomes == plain.chromosomes == ["chr1", "chr2"]
    np.testing.assert_array_equal(fancy.col_pos, plain.col_pos)
    assert list(plain.col_pos) == [2, 5, -1, 3, 0, 4, -1, 1]  # chr1: 10, 20, 30, NaN; chr2: 1, 5
    assert fancy.n_without_position == 1
    plain.close()
    fancy.close()


@pytest.mark.parametrize("method", ["ward", "single", "average"])
@pytest.mark.parametrize("n", [2, 3, 17, 200])
def test_leaves_list_matches_scipy(n, method):
    """Host side of `pl.chromosome_heatmap(dendrogram=True)` / `tl.cell_linkage`: the leaf order read off a scipy
    style linkage matrix (iterative, left child first) is scipy's."""
    from scipy.cluster.hierarchy import leaves_list as scipy_leaves
    from scipy.cluster.hierarchy import linkage

    from infercnvpy_amd.tl._linkage import leaves_list

    rng = np.random.default_rng(n)
    Z = linkage(rng.normal(size=(n, 5)), method=method)
    np.testing.assert_array_equal(leaves_list(Z), scipy_leaves(Z))


def _toy_adata(names=("a", "b", "c"), **var_cols):
    from infercnvpy_amd._compat import SimpleAnnData

    cols = {"chromosome": ["chr1"] * len(names), "start": list(range(len(names))),
            "end": [s + 10 for s in range(len(names))]}
    cols.update(var_cols)
    cols = {k: v for k, v in cols.items() if v is not None}
    var = pd.DataFrame(cols, index=pd.Index(list(names)))
    return SimpleAnnData(np.ones((4, len(names)), dtype=np.float32), var=var)


def test_public_api_validation_errors_come_before_any_gpu_work():
    """The reference raises ValueError for duplicate var_names (tl/_infercnv.py:97-98) and for missing genomic
    position columns (:99-102) before it touches the matrix; so does the drop-in (no GPU in this test)."""
    import infercnvpy_amd as cnv

    with pytest.raises(ValueError, match="unique"):
        cnv.tl.infercnv(_toy_adata(names=("a", "a", "b")))
    for missing in ("chromosome", "start", "end"):
        with pytest.raises(ValueError, match="Genomic positions"):
            cnv.tl.infercnv(_toy_adata(**{missing: None}))


def test_cnv_score_argument_errors_come_before_any_gpu_work():
    """`tl.cnv_score`: ValueError when the default groupby is absent (reference tl/_scores.py:63-64); the
    deprecated `obs_key` spelling warns (:54-61) and then takes the same route."""
    import infercnvpy_amd as cnv

    ad = _toy_adata()
    with pytest.raises(ValueError, match="cnv_leiden"):
        cnv.tl.cnv_score(ad)
    with pytest.warns(FutureWarning), pytest.raises(ValueError, match="cnv_leiden"):
        cnv.tl.cnv_score(ad, obs_key="cnv_leiden")


def test_missing_extension_fails_loudly_no_cpu_fallback(tmp_path):
    """No CPU fallback: with the shared library absent the public entry point raises HipExtensionMissing (a
    RuntimeError) instead of computing anything.  Fresh interpreter, library path pointed at nothing."""
    import subprocess
    import sys

    code = (
        "import numpy as np, pandas as pd\n"
        "import infercnvpy_amd as cnv\n"
        "from infercnvpy_amd._compat import SimpleAnnData\n"
        "from infercnvpy_amd._lib import HipExtensionMissing\n"
        "var = pd.DataFrame({'chromosome': ['chr1'] * 3, 'start': [1, 2, 3], 'end': [2, 3, 4]}, index=list('abc'))\n"
        "ad = SimpleAnnData(np.ones((2, 3), dtype=np.float32), var=var)\n"
        "try:\n"
        "    cnv.tl.infercnv(ad, window_size=2, step=1)\n"
        "except HipExtensionMissing as e:\n"
        "    assert isinstance(e, RuntimeError) and 'no CPU fallback' in str(e)\n"
        "    print('LOUD')\n"
    )
    env = dict(os.environ, INFERCNV_HIP_LIB=str(tmp_path / "absent.so"), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LOUD" in r.stdout, r.stderr[-2000:]
    assert "X_cnv" not in r.stdout


def test_product_package_never_imports_the_oracle():
    """`oracle/` is test infrastructure: only tests, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg
    may import it.  Static check of every module of the shipped package (and of the C sources' includes)."""
    import ast

    pkg = os.path.join(ROOT, "infercnvpy_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(d, f)
            if f.endswith(".py"):
                for node in ast.walk(ast.parse(open(path).read())):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or ""]
                    if any(n.split(".")[0] == "oracle" for n in names):
                        offenders.append(path)
            elif f.endswith((".hip", ".hpp", ".h")):
                if re.search(r'#include\s+[<"][^>"]*oracle', open(path).read()):
                    offenders.append(path)
    assert not offenders, offenders
    # bench.py: the oracle appears only in the cpu_baseline leg (cpu_baseline() and its pool workers)
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle" for n in ast.walk(fn))
        assert not uses or fn.name in ("cpu_baseline", "_cpu_worker", "_cpu_chunk_worker"), fn.name
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n) for n in tree.body)


@pytest.mark.parametrize("genes,window,step", [
    ([57, 23, 9], 10, 3), ([120, 40], 25, 5), ([64, 64, 5], 8, 4), ([33], 11, 1), ([200, 90, 12], 100, 10),
    ([75, 31], 6, 7),  # step > window: genes between two windows have no value
])
def test_gene_run_tables_are_the_oracles_window_coverage(genes, window, step):
    """``calculate_gene_values`` on the CPU side: the plan's runs (``icv_plan_gene_runs``: which windows a gene's value
    averages) against the oracle's restatement of ``_calculate_gene_averages`` (reference tl/_infercnv.py:247-298) --
    a one-hot window row through ``gene_values_from_windows`` shows exactly which genes window j reaches, and with
    which 1 / count."""
    v = cases.synthetic_var(genes, extra=(("chrX", 7), (None, 3)))
    chrom, start = v["chromosome"], v["start"]
    plan = GenePlan(chrom, start, window_size=window, step=step)
    runs = plan.gene_runs()
    n_cols = len(chrom)
    assert runs["col_run"].shape == (n_cols,) and runs["genes"].sum() == (runs["col_run"] >= 0).sum()
    assert (np.bincount(runs["col_run"][runs["col_run"] >= 0], minlength=len(runs["genes"])) == runs["genes"]).all()
    keep = ~plan.var_mask
    ch, st = chrom[keep], start[keep]
    kept_idx = np.flatnonzero(keep)
    covered = np.zeros(n_cols, dtype=bool)
    for c in O.used_chromosomes(ch):
        cols = kept_idx[O.chromosome_gene_order(ch, st, c)]  # input columns of the chromosome, position order
        g = len(cols)
        w0 = int(plan.chr_pos[c])
        n_w = O.smooth_segment(np.zeros((1, g)), window, step).shape[1]
        cover = O.gene_values_from_windows(np.eye(n_w), g, window, step)  # [window j, gene p] = 1 / count or 0 / NaN
        for p, col in enumerate(cols):
            js = np.flatnonzero(np.nan_to_num(cover[:, p]) > 0)
            r = runs["col_run"][col]
            if js.size == 0:
                assert r == -1 and np.isnan(cover[:, p]).all()
                continue
            covered[col] = True
            assert r >= 0
            assert runs["first"][r] == w0 + js[0] and runs["count"][r] == js.size
            assert (np.diff(js) == 1).all() and np.allclose(cover[js, p], 1.0 / js.size)
    assert ((runs["col_run"] >= 0) == covered).all()  # masked chromosomes / null positions: no value
    # runs are maximal: neighbours differ
    same = (runs["first"][1:] == runs["first"][:-1]) & (runs["count"][1:] == runs["count"][:-1])
    assert not same.any()
    plan.close()

"""Pin the CPU oracle (oracle/infercnv_oracle.py) to the reference.

(1) the reference's own known-answer tests, transcribed as arrays
    (reference tests/test_tools.py:11-39, 64-191; tests/conftest.py:42-139;
    tests/test_scores.py:18-21);
(2) vectors captured by running the reference in the build container
    (tests/golden/make_golden.py) -- bit-exact float64 comparison.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import cases
from _golden import GoldenCase, case_names
from oracle import infercnv_oracle as O


# ---- (1) transcribed known-answer tests ------------------------------------
def test_running_mean_n_less_than_genes():
    x = np.array([[1, 2, 3, 4, 5], [6, 7, 8, 9, 10]])
    np.testing.assert_array_equal(O.smooth_segment(x, 3, 1), np.array([[2, 3, 4], [7, 8, 9]]))
    gv = O.gene_values_from_windows(O.smooth_segment(x, 3, 1), 5, 3, 1)
    np.testing.assert_array_equal(gv, np.array([[2.0, 2.5, 3.0, 3.5, 4.0], [7.0, 7.5, 8.0, 8.5, 9.0]]))


def test_running_mean_n_greater_than_genes():
    x = np.array([[1, 2, 3, 4, 5], [6, 7, 8, 9, 10]])
    sm = O.smooth_segment(x, 7, 1)
    np.testing.assert_array_equal(sm, np.array([[3], [8]]))
    np.testing.assert_array_equal(O.gene_values_from_windows(sm, 5, 7, 1), np.repeat(sm, 5, axis=1))


def test_gene_averages():
    sm = np.array([[2, 3, 4], [4, 4, 6], [6, 2, 1]], dtype=float)
    gv = O.gene_values_from_windows(sm, 5, 3, 1)
    exp = np.array([[2.0, 2.5, 3.0, 3.5, 4.0], [4.0, 4.0, 4.666667, 5.0, 6.0], [6.0, 4.0, 3.0, 1.5, 1.0]])
    np.testing.assert_allclose(gv, exp, atol=1e-6)


X_RES_ACTUAL = np.array(
    [
        [1.00, 0.00, 0.00, 0.00, 0.00, 1.00],
        [-1.00, 0.00, 0.00, 0.00, 0.00, 0.00],
        [0.00, 1.25, 1.25, 0.00, 0.00, 0.00],
        [0.00, 0.00, 0.00, 0.875, 0.00, 0.00],
    ]
)
GENE_RES_ACTUAL = np.array(
    [
        [0.75, 0.00, 0.000000, 0.00, -0.75, 0.000000, 0.000000, 0.0, 0.0, 0.75],
        [-1.00, 0.00, 0.000000, 0.00, 0.00, 0.000000, 0.000000, 0.0, 0.0, 0.00],
        [0.00, 0.75, 0.91666667, 1.25, 1.25, 0.000000, 0.000000, 0.0, 0.0, 0.00],
        [0.00, 0.00, 0.000000, 0.00, 0.00, 0.921875, 0.703125, 0.0, 0.0, 0.00],
    ]
)


@pytest.mark.parametrize("with_genes", [False, True])
def test_chunk_kernel_on_reference_fixture(with_genes):
    x, chrom, start, _ = cases.adata_full_mock_arrays()
    X = sp.csr_matrix(x)
    ref = O.reference_profile(X, None, None, None, 10)
    chr_pos, res, gene_res, _ = O.infercnv_chunk(X, chrom, start, ref, 1, 3, 1, 1, calculate_gene_values=with_genes)
    np.testing.assert_array_equal(res, X_RES_ACTUAL)
    assert chr_pos == {"chr1": 0, "chr2": 3}
    if with_genes:
        cols, vals = gene_res
        full = np.full((4, 10), np.nan)
        full[:, cols] = vals
        np.testing.assert_allclose(full, GENE_RES_ACTUAL, atol=1e-8)
    else:
        assert gene_res is None


def test_driver_more_than_2_chunks():
    x, chrom, start, _ = cases.adata_full_mock_arrays()
    chr_pos, res, per_gene, _ = O.infercnv(
        sp.csr_matrix(x), chrom, start, chunksize=2, lfc_clip=1, window_size=3, step=1, dynamic_threshold=1,
        calculate_gene_values=True,
    )
    np.testing.assert_array_equal(per_gene[0], np.array([0.75, 0.0, 0.0, 0.0, -0.75, 0.0, 0.0, 0.0, 0.0, 0.75]))
    np.testing.assert_array_equal(per_gene[3], np.array([0, 0, 0, 0, 0, 0.921875, 0.703125, 0, 0, 0]))
    np.testing.assert_array_equal(res.toarray(), X_RES_ACTUAL)
    assert chr_pos == {"chr1": 0, "chr2": 3}


ADATA_MOCK_X = np.array([[1, 1, 1, 2], [2, 1, 2, 2], [5, 5, 5, 5], [7, 5, 5, 7], [9, 9, 9, 9]])
ADATA_MOCK_CAT = np.array(["foo", "foo", "bar", "baz", "bar"])


@pytest.mark.parametrize("wrap", [np.array, sp.csr_matrix, sp.csc_matrix])
def test_reference_profile(wrap):
    X = wrap(ADATA_MOCK_X)
    np.testing.assert_almost_equal(
        np.asarray(O.reference_profile(X, ADATA_MOCK_CAT, ["foo", "baz"], None, 4)),
        np.array([[1.5, 1, 1.5, 2], [7, 5, 5, 7]]),
    )
    np.testing.assert_almost_equal(
        np.asarray(O.reference_profile(X, None, None, None, 4)), np.array([[4.8, 4.2, 4.4, 5]]), decimal=5
    )
    given = np.array([1, 2, 3, 4])
    np.testing.assert_equal(given, O.reference_profile(X, ADATA_MOCK_CAT, "bar", given, 4)[0, :])
    with pytest.raises(ValueError):
        O.reference_profile(X, ADATA_MOCK_CAT, "bar", np.array([1, 2, 3]), 4)
    with pytest.raises(ValueError):
        O.reference_profile(X, ADATA_MOCK_CAT, ["foo", "nope"], None, 4)


@pytest.mark.parametrize("wrap", [np.array, sp.csr_matrix, sp.csc_matrix])
def test_cnv_score_known_answer(wrap):
    x_cnv = wrap(np.array([[1, 1, 1, 2, 2, 1, 1, 1], [2, 2, 2, 1, 1, 2, 2, 2], [4, 4, 4, 2, 2, 3, 3, 3],
                           [2, 2, 2, 4, 4, 4, 4, 4]]).T)
    res = O.cnv_score(x_cnv, list("AAAAABBB"))
    assert res["A"] == pytest.approx(2.25, abs=0.001)
    assert res["B"] == pytest.approx(2.5, abs=0.001)


def test_natural_order():
    assert O.natural_order(["chr10", "chr2", "chrX", "chr1", "chr22", "chr3"]) == [
        "chr1", "chr2", "chr3", "chr10", "chr22", "chrX"]


# ---- (2) captured reference vectors ----------------------------------------
@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_captured_reference(name):
    g = GoldenCase(name)
    chr_pos, res, per_gene, _ = O.infercnv(g.X, g.chromosome, g.start, **g.array_kwargs())
    assert {k: int(v) for k, v in chr_pos.items()} == g.chr_pos
    assert list(chr_pos.keys()) == list(g.chr_pos.keys())
    np.testing.assert_array_equal(res.toarray(), g.out)
    if g.per_gene is not None:
        np.testing.assert_array_equal(np.isnan(per_gene), np.isnan(g.per_gene))
        np.testing.assert_array_equal(np.nan_to_num(per_gene), np.nan_to_num(g.per_gene))


@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_oracle_reference_means_match_captured(fmt):
    import os

    from _golden import GOLDEN_DIR

    z = np.load(os.path.join(GOLDEN_DIR, f"refmean_{fmt}.npz"))
    X = sp.csr_matrix(z["X"]) if fmt == "csr" else z["X"]
    np.testing.assert_array_equal(np.asarray(O.reference_profile(X, None, None, None, X.shape[1])), z["r_all"])
    np.testing.assert_array_equal(
        np.asarray(O.reference_profile(X, z["obs"], ["normalA", "normalB"], None, X.shape[1])), z["r_cat"])


# --------------------------------------------------------------------------- #
# ithgex / ithcna: the reference's own known answers (tests/test_scores.py:6-15, conftest.py:111-138)
# --------------------------------------------------------------------------- #
ITH_X = np.array([[1, 1, 1, 1, 1, 1, 2, 3], [2, 2, 2, 2, 2, 2, 8, 0], [3, 3, 3, 3, 3, 10, 3, 7]]).T
ITH_CNV = np.array(
    [[1, 1, 1, 2, 2, 1, 1, 1], [2, 2, 2, 1, 1, 2, 2, 2], [4, 4, 4, 2, 2, 3, 3, 3], [2, 2, 2, 4, 4, 4, 4, 4]]
).T
ITH_GROUPS = list("AAAAABBB")


@pytest.mark.parametrize("fmt", [np.array, sp.csr_matrix, sp.csc_matrix])
def test_ith_score_reference_known_answers(fmt):
    gex = O.ith_score(fmt(ITH_X), ITH_GROUPS)
    assert gex["A"] == 0
    assert gex["B"] == pytest.approx(1.2628, abs=0.001)
    cna = O.ith_score(fmt(ITH_CNV), ITH_GROUPS)
    assert cna["A"] == pytest.approx(1.053, abs=0.001)
    assert cna["B"] == 0


def test_ith_score_captured_reference():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ith_scores.npz"), allow_pickle=False)
    X, labels = z["X"], z["labels"]
    gex = O.ith_score(X, labels)
    cna = O.ith_score(sp.csr_matrix(np.where(np.abs(X) > 0.8, X, 0).astype(np.float64)), labels)
    assert sorted(gex) == list(z["keys"])  # the single-cell group is skipped
    for i, g in enumerate(z["keys"]):
        np.testing.assert_array_equal(gex[g], z["gex"][i])
        np.testing.assert_array_equal(cna[g], z["cna"][i])  # includes a NaN score (constant profile)

"""CPU tests of the host logic of the multi-GPU ``tl.infercnv``: which GPU gets which rows, how the shards'
reference sums become means, how the shards' CSR results are concatenated (reference: ``process_map`` over chunks +
``vstack``, ``tl/_infercnv.py:120-137``; ``_get_reference`` :359-408)."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from infercnvpy_amd.dist import shard_bounds
from infercnvpy_amd.tl import _infercnv as T


class _FakeCuda:
    def __init__(self, n, cur=0):
        self.n, self.cur = n, cur

    def device_count(self):
        return self.n

    def current_device(self):
        return self.cur


class _FakeTorch:
    def __init__(self, n, cur=0):
        self.cuda = _FakeCuda(n, cur)

    @staticmethod
    def device(d):
        import torch

        return torch.device(d)


def test_resolve_devices():
    t8 = _FakeTorch(8, cur=3)
    # explicit list wins, repeats allowed, capped by the number of chunks
    assert T._resolve_devices(None, [0, 0, 5], 100, t8) == [0, 0, 5]
    assert T._resolve_devices(2, [1, 2, 3, 4], 2, t8) == [1, 2]
    assert T._resolve_devices(None, ["cuda:2"], 10, t8) == [2]
    with pytest.raises(ValueError):
        T._resolve_devices(None, [8], 10, t8)
    with pytest.raises(ValueError):
        T._resolve_devices(None, [], 10, t8)
    # n_jobs = the reference's worker count: k GPUs, the caller's current device first, capped by what exists
    assert T._resolve_devices(4, None, 100, t8) == [3, 0, 1, 2]
    assert T._resolve_devices(64, None, 100, t8) == [3, 0, 1, 2, 4, 5, 6, 7]
    assert T._resolve_devices(1, None, 100, t8) == [3]
    assert T._resolve_devices(4, None, 100, _FakeTorch(1)) == [0]
    # None: every GPU that gets at least four chunks
    assert T._resolve_devices(None, None, 200, t8) == [3, 0, 1, 2, 4, 5, 6, 7]
    assert T._resolve_devices(None, None, 9, t8) == [3, 0]
    assert T._resolve_devices(None, None, 3, t8) == [3]
    assert T._resolve_devices(None, None, 0, t8) == [3]
    # ... and 50 000 cells
    assert T._resolve_devices(None, None, 200, t8, n_obs=1_000_000) == [3, 0, 1, 2, 4, 5, 6, 7]
    assert T._resolve_devices(None, None, 200, t8, n_obs=120_000) == [3, 0]
    assert T._resolve_devices(None, None, 200, t8, n_obs=4000) == [3]


def _chain_dense(X, rows, acc):
    """CPU restatement of icv_colchain on a dense matrix: one sequential chain per column, in the matrix dtype."""
    for r in rows:
        acc = acc + X[r]
    return acc


def _chain_csr(S, rows, scale, acc):
    """... on a CSR matrix: fl(x * scale) added row after row (scipy: (X * (1 / n)).sum(axis=0) = ones @ X)."""
    for r in rows:
        a, b = S.indptr[r], S.indptr[r + 1]
        acc[S.indices[a:b]] = acc[S.indices[a:b]] + S.data[a:b] * scale
    return acc


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64])
def test_means_from_shard_chains_equal_the_reference_bit_for_bit(dtype):
    """The evaluation order the GPU path reproduces (csrc/icv_kernel_chain.hpp), restated with plain loops on the CPU
    and chained over three row shards as tl.infercnv chains them, against the oracle's reference_profile = the numpy /
    scipy calls of the reference (:385, :400): array_equal, dense and CSR, all cells and categories (one listed twice)."""
    import scipy.sparse as sp

    from oracle import infercnv_oracle as O

    rng = np.random.RandomState(0)
    X = rng.gamma(0.3, 1.0, size=(230, 40))
    X[X < 0.5] = 0
    X = (X * 3).astype(dtype)
    acc_dtype = np.float32 if dtype == np.float32 else np.float64
    labels = np.array(["a", "b", "c"])[rng.randint(0, 3, 230)]
    obs = pd.DataFrame({"g": labels})
    bounds = shard_bounds(230, 3, 50)
    S = sp.csr_matrix(X)
    # all-cell mean
    acc = np.zeros(40, acc_dtype)
    acc_s = np.zeros(40, acc_dtype)
    for a, b in bounds:  # shard k continues shard k - 1's accumulators
        acc = _chain_dense(X.astype(acc_dtype), range(a, b), acc)
        acc_s = _chain_csr(S.astype(acc_dtype), range(a, b), acc_dtype(1.0 / 230), acc_s)
    ref = T._means_from_chains([acc], 230.0, None, False)
    np.testing.assert_array_equal(ref, np.asarray(O.reference_profile(X, None, None, None, 40)))
    ref = T._means_from_chains([acc_s], 230.0, None, True)
    np.testing.assert_array_equal(ref, np.asarray(O.reference_profile(S, None, None, None, 40)))
    assert ref.dtype == acc_dtype
    # per category, a category listed twice repeats its row (np.isin semantics of the reference)
    groups, counts, cats = T._reference_groups(obs, "g", ["b", "a", "b"])
    assert counts.tolist() == [(labels == "b").sum(), (labels == "a").sum(), (labels == "b").sum()]
    assert set(np.unique(groups)) == {-1, 0, 1}  # the second "b" never owns rows
    accs = [np.zeros(40, acc_dtype) for _ in range(3)]
    accs_s = [np.zeros(40, acc_dtype) for _ in range(3)]
    for a, b in bounds:
        for gi in range(3):
            rows = a + np.nonzero(groups[a:b] == gi)[0]
            accs[gi] = _chain_dense(X.astype(acc_dtype), rows, accs[gi])
            accs_s[gi] = _chain_csr(S.astype(acc_dtype), rows, acc_dtype(1.0 / counts[gi]), accs_s[gi])
    ref = T._means_from_chains(accs, counts, cats, False)
    np.testing.assert_array_equal(ref, np.asarray(O.reference_profile(X, labels, ["b", "a", "b"], None, 40)))
    ref = T._means_from_chains(accs_s, counts, cats, True)
    np.testing.assert_array_equal(ref, np.asarray(O.reference_profile(S, labels, ["b", "a", "b"], None, 40)))
    groups, counts, cats = T._reference_groups(obs, "g", "c")
    assert counts.tolist() == [(labels == "c").sum()] and (groups >= 0).sum() == counts[0]
    with pytest.raises(ValueError):
        T._reference_groups(obs, "g", ["a", "nope"])


def test_concat_csr_is_vstack():
    rng = np.random.RandomState(1)
    mats = [sp.random(r, 37, density=d, random_state=rng, format="csr", dtype=np.float64)
            for r, d in ((50, 0.2), (1, 0.0), (70, 0.5), (13, 1.0))]
    parts = []
    for m in mats:
        cap = m.nnz + 5  # the drains hand over arrays with spare capacity: only [:nnz] counts
        ix = np.zeros(cap, dtype=np.int32)
        dv = np.zeros(cap, dtype=np.float64)
        ix[: m.nnz], dv[: m.nnz] = m.indices, m.data
        parts.append((m.indptr.astype(np.int64), ix[: m.nnz], dv[: m.nnz]))
    got = T._concat_csr(parts, sum(m.shape[0] for m in mats), 37)
    exp = sp.vstack(mats).tocsr()
    assert got.shape == exp.shape
    np.testing.assert_array_equal(got.indptr, exp.indptr)
    np.testing.assert_array_equal(got.indices, exp.indices)
    np.testing.assert_array_equal(got.data, exp.data)


def _numpy_pairwise(a):
    """numpy's pairwise summation (8 accumulators up to 128 elements, halves rounded down to multiples of 8 above),
    restated: what k_colpair_csc walks on the GPU."""
    n = len(a)
    if n < 8:
        res = a.dtype.type(0)
        for v in a:
            res = res + v
        return res
    if n <= 128:
        r = [a[k] for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = r[k] + a[i + k]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _numpy_pairwise(a[:n2]) + _numpy_pairwise(a[n2:])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_csc_mean_is_first_entry_plus_numpy_pairwise_of_the_rest(dtype):
    """scipy reduces a CSC matrix over axis 0 with np.add.reduceat per column (after scaling by 1 / n): the order
    icv_colmean_csc restates, here in plain Python against the oracle (= scipy) on columns of 0 .. 3000 entries."""
    from oracle import infercnv_oracle as O

    rng = np.random.RandomState(1)
    n, g = 3000, 24
    X = rng.gamma(0.3, 1.0, size=(n, g)).astype(dtype)
    for c, dens in enumerate(np.linspace(0, 1, g)):
        X[rng.rand(n) >= dens, c] = 0
    X[:, -1] = rng.gamma(0.3, 1.0, n) + 0.1
    S = sp.csc_matrix(X)
    labels = np.array(["a", "b"])[rng.randint(0, 2, n)]
    for sel, cats in ((np.ones(n, bool), None), (labels == "b", ["b"])):
        inv = dtype(1.0 / sel.sum())
        out = np.zeros(g, dtype)
        for c in range(g):
            a, b = S.indptr[c], S.indptr[c + 1]
            p = (S.data[a:b] * inv)[sel[S.indices[a:b]]]
            if len(p):
                out[c] = p[0] + _numpy_pairwise(p[1:]) if len(p) > 1 else p[0]
        exp = np.asarray(O.reference_profile(S, labels if cats else None, cats, None, g))[0]
        np.testing.assert_array_equal(out, exp)

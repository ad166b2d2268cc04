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
    # n_jobs = the reference's worker count: k GPUs, capped by what exists; 1 = the current device
    assert T._resolve_devices(4, None, 100, t8) == [0, 1, 2, 3]
    assert T._resolve_devices(64, None, 100, t8) == list(range(8))
    assert T._resolve_devices(1, None, 100, t8) == [3]
    assert T._resolve_devices(4, None, 100, _FakeTorch(1)) == [0]
    # None: every GPU that gets at least four chunks
    assert T._resolve_devices(None, None, 200, t8) == list(range(8))
    assert T._resolve_devices(None, None, 9, t8) == [0, 1]
    assert T._resolve_devices(None, None, 3, t8) == [3]
    assert T._resolve_devices(None, None, 0, t8) == [3]
    # ... and 50 000 cells
    assert T._resolve_devices(None, None, 200, t8, n_obs=1_000_000) == list(range(8))
    assert T._resolve_devices(None, None, 200, t8, n_obs=120_000) == [0, 1]
    assert T._resolve_devices(None, None, 200, t8, n_obs=4000) == [3]


def test_means_from_shard_sums_equal_the_reference_semantics():
    rng = np.random.RandomState(0)
    X = rng.gamma(0.3, 1.0, size=(230, 40)).astype(np.float32)
    labels = np.array(["a", "b", "c"])[rng.randint(0, 3, 230)]
    obs = pd.DataFrame({"g": labels})
    # all-cell mean
    bounds = shard_bounds(230, 3, 50)
    total = sum(X[a:b].sum(axis=0, dtype=np.float64)[None, :] for a, b in bounds)
    ref = T._means_from_sums(total, 230.0, None, np.float32)
    np.testing.assert_array_equal(ref, (X.sum(axis=0, dtype=np.float64) / 230).astype(np.float32)[None, :])
    # per category, a category listed twice repeats its row (np.isin semantics of the reference)
    groups, counts, cats = T._reference_groups(obs, "g", ["b", "a", "b"])
    assert counts.tolist() == [(labels == "b").sum(), (labels == "a").sum(), (labels == "b").sum()]
    assert set(np.unique(groups)) == {-1, 0, 1}  # the second "b" never owns rows
    sums = np.vstack([X[groups == k].sum(axis=0, dtype=np.float64) for k in range(3)])
    ref = T._means_from_sums(sums, counts, cats, np.float64)
    exp = np.vstack([X[labels == c].sum(axis=0, dtype=np.float64) / (labels == c).sum() for c in ["b", "a", "b"]])
    np.testing.assert_array_equal(ref, exp)
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

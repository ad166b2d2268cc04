"""The block-wise exact float32 chain (tests/exact_chain_proto.py) against numpy / scipy: ``array_equal`` to the
reference's own reductions (``np.mean(X, axis=0)`` of a C-ordered dense matrix, scipy's CSR mean: reference
``tl/_infercnv.py:385, :400``) on the means fuzzer's generator and on inputs built to hit every replay cause."""
import numpy as np
import pytest
import scipy.sparse as sp

import exact_chain_proto as P


def _numpy_chain(X, start=None):
    """numpy's own order for a C-contiguous matrix: one chain per column over the rows, in float32."""
    s = np.zeros(X.shape[1], dtype=np.float32) if start is None else np.array(start, dtype=np.float32)
    for r in range(X.shape[0]):
        s = (s + X[r]).astype(np.float32)
    return s


def _case(seed):
    rs = np.random.RandomState(seed)
    n = int(rs.choice([1, 2, 63, 64, 65, 127, 129, 500, 1000, 3000]) if rs.rand() < 0.5 else rs.randint(1, 3001))
    g = int(rs.choice([1, 5, 31, 96, 128, 400]))
    dens = float(10 ** rs.uniform(-3, 0)) if rs.rand() < 0.8 else 1.0
    X = rs.gamma(0.3, 1.0, (n, g)).astype(np.float32)
    X[rs.rand(n, g) > dens] = 0
    kind = rs.randint(6)
    if kind == 1:
        X = np.floor(X * 4).astype(np.float32)  # integer counts: exact sums, no rounding at all below 2^24
    elif kind == 2:
        X = (np.round(X * 8) / 8 + (X > 0) * 1024).astype(np.float32)  # few distinct values, large sums: ties
    elif kind == 3:
        X = (X * np.float32(10.0) ** rs.randint(-30, 30, size=g).astype(np.float32)).astype(np.float32)  # column scales
    elif kind == 4 and n > 3:
        X[rs.randint(n), rs.randint(g)] = -1.5  # a negative entry: its block is replayed
    return X, int(rs.choice([1, 7, 64, 256]))


@pytest.mark.parametrize("seed", range(60))
def test_dense_chain_is_numpys_bits(seed):
    X, block = _case(seed)
    st = {}
    got = P.chain_by_blocks(X, block=block, stats=st)
    np.testing.assert_array_equal(got.view(np.int32), _numpy_chain(X).view(np.int32))
    if X.shape[0] >= 64 and X.shape[1] > 1:
        # the reference's public reduction itself (numpy adds a C-contiguous matrix row by row in the matrix dtype; a
        # single-column matrix is a 1-D contiguous reduction for numpy: its pairwise order)
        with np.errstate(over="ignore", invalid="ignore"):
            np.testing.assert_array_equal(got.view(np.int32), np.add.reduce(X, axis=0).view(np.int32))


@pytest.mark.parametrize("seed", range(100, 130))
def test_csr_mean_is_scipys_bits(seed):
    X, block = _case(seed)
    X = np.abs(X)
    Xc = sp.csr_matrix(X)
    Xs = P.csr_scaled_dense(Xc, X.shape[0])
    got = P.chain_by_blocks(Xs.toarray(), block=block)
    exp = np.asarray(Xc.mean(axis=0)).ravel()
    assert exp.dtype == np.float32
    np.testing.assert_array_equal(got.view(np.int32), exp.view(np.int32))


def test_every_replay_cause_is_exercised_and_counted():
    rs = np.random.RandomState(5)
    X = rs.gamma(0.3, 1.0, (4000, 64)).astype(np.float32)
    X[X < 0.5] = 0
    st = {}
    got = P.chain_by_blocks(X, block=64, stats=st)
    np.testing.assert_array_equal(got.view(np.int32), _numpy_chain(X).view(np.int32))
    assert st["no_start"] > 0 and st["crossing"] > 0 and 0 < st["replayed"] < 0.6 * st["blocks"]
    # ties: halves added to a sum that is a multiple of 2 ulps apart -- every row ties in the binade [2^24, 2^25)
    T = np.full((300, 4), 1.0, dtype=np.float32)
    T[0] = 2.0 ** 24
    st2 = {}
    got = P.chain_by_blocks(T, block=32, stats=st2)
    np.testing.assert_array_equal(got.view(np.int32), _numpy_chain(T).view(np.int32))
    assert st2["ties"] > 0
    assert float(got[0]) == 2.0 ** 24  # round-half-even: 2^24 + 1 ties back to 2^24 every time


def test_continuation_and_wrong_estimates_stay_exact():
    """A chain continued from another shard's exact values with a float64 ESTIMATE of that start (the sharded form):
    whatever the estimate says -- right, one binade off, nonsense -- the result is the chain's; a wrong estimate only
    costs replays."""
    rs = np.random.RandomState(11)
    A = rs.gamma(0.3, 1.0, (1500, 40)).astype(np.float32)
    B = rs.gamma(0.3, 1.0, (1700, 40)).astype(np.float32)
    whole = _numpy_chain(np.vstack([A, B]))
    sA = P.chain_by_blocks(A)
    est = A.sum(axis=0, dtype=np.float64)
    for e in (est, est * 2.1, est * 0.3, np.zeros(40), np.full(40, np.nan)):
        st = {}
        got = P.chain_by_blocks(B, start=sA, estimate_start=e, stats=st)
        np.testing.assert_array_equal(got.view(np.int32), whole.view(np.int32))
    # the two halves of a rank: records from the estimate (concurrent), then the scan from the exact hand-over
    recs = P.rank_records(B, est, block=128)
    np.testing.assert_array_equal(P.rank_scan(B, sA, recs, block=128).view(np.int32), whole.view(np.int32))

"""GPU tests (``-m gpu``) of the reference-order column means (``icv_colchain`` / ``icv_colmean_csc``).

The reference's default call (``reference=None`` / ``reference_cat``) forms its profile with ``np.mean(X, axis=0)``
(reference tl/_infercnv.py:385, :400).  numpy / scipy evaluate that in a fixed order -- sequential float32 chains per
column for a C-contiguous matrix and for CSR, ``np.add.reduceat`` per column for CSC -- and a float32 sum is not
associative, so "the same mean" means "the same order".  Every comparison here is ``array_equal`` (bit for bit) against
the oracle's ``reference_profile``, which calls the very numpy / scipy primitives the reference calls.
"""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _expr(n, g, seed, dtype=np.float32, density=None):
    rs = np.random.RandomState(seed)
    X = rs.gamma(0.3, 1.0, size=(n, g)).astype(dtype)
    if density is None:
        X[X < 0.5] = 0
    else:
        X[rs.rand(n, g) >= density] = 0
    return X


def _oracle_means(X, labels=None, cats=None):
    from oracle import infercnv_oracle as O

    return np.asarray(O.reference_profile(X, labels, cats, None, X.shape[1]))


def _gpu_means(X, labels=None, cats=None, pieces=1):
    """The engine's calls as tl.infercnv makes them: row pieces in order, one chain per category."""
    from infercnvpy_amd import _engine

    torch = _engine._torch()
    n = X.shape[0]
    Xd = X.astype(np.float64) if X.dtype.kind in "iub" else X
    dm = _engine.to_device_matrix(Xd)
    groups = [None] if cats is None else [np.asarray(labels) == c for c in cats]
    bounds = np.linspace(0, n, pieces + 1).astype(int)
    out = []
    for sel in groups:
        count = n if sel is None else int(sel.sum())
        acc = None
        for r0, r1 in zip(bounds[:-1], bounds[1:]):
            if r1 == r0:
                continue
            rows = None if sel is None else np.nonzero(sel[r0:r1])[0]
            acc = _engine.column_chain(dm, acc, rows, count, int(r0), int(r1))
        out.append(_engine.chain_mean(acc, count, sp.issparse(X)).cpu().numpy())
    torch.cuda.synchronize()
    return np.vstack(out)


DENSE_SHAPES = [(3001, 1337), (999, 20003), (4000, 20000), (61, 130), (1, 3), (29, 5), (2500, 40001), (700, 257)]


@pytest.mark.parametrize("shape", DENSE_SHAPES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dense_all_cells_mean_is_numpy_bit_for_bit(shape, dtype):
    X = _expr(*shape, seed=shape[0], dtype=dtype)
    got = _gpu_means(X)
    exp = _oracle_means(X)
    assert got.dtype == exp.dtype
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("pieces", [1, 3, 7])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64])
def test_dense_category_means_over_row_pieces(pieces, dtype):
    n, g = 2311, 5003
    X = _expr(n, g, seed=5) if dtype != np.int64 else np.random.RandomState(5).poisson(0.4, (n, g)).astype(np.int64)
    X = X.astype(dtype)
    labels = np.array(["a", "b", "c", "d"])[np.random.RandomState(6).randint(0, 4, n)]
    got = _gpu_means(X, labels, ["c", "a"], pieces=pieces)
    exp = _oracle_means(X, labels, ["c", "a"])
    np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(_gpu_means(X, pieces=pieces), _oracle_means(X))


@pytest.mark.parametrize("shape,density", [((5000, 20000), 0.07), ((1200, 20003), 0.3), ((300, 700), 0.02),
                                            ((777, 40001), 0.05), ((64, 129), 1.0), ((2000, 5000), 0.0005)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_csr_means_are_scipy_bit_for_bit(shape, density, dtype):
    X = sp.csr_matrix(_expr(*shape, seed=shape[1], dtype=dtype, density=density))
    np.testing.assert_array_equal(_gpu_means(X), _oracle_means(X))
    labels = np.array(["n", "t", "u"])[np.random.RandomState(1).randint(0, 3, shape[0])]
    for pieces in (1, 4):
        got = _gpu_means(X, labels, ["t", "n"], pieces=pieces)
        np.testing.assert_array_equal(got, _oracle_means(X, labels, ["t", "n"]))


def test_csr_chain_routes_agree(monkeypatch):
    """The per-column-queue kernel (k_colchain_csrq), its guarded-load path for rows too far apart for a buffer offset
    (forced with a small ICV_CHAIN_FAR), the round-4 kernel (ICV_NO_CHAIN_QUEUES, also what matrices wider than 65 535
    columns take) -- all scipy's bits."""
    from infercnvpy_amd import _lib

    lib = _lib.load()
    X = sp.csr_matrix(_expr(1500, 3001, seed=12, density=0.1))
    labels = np.array(["n", "t", "u"])[np.random.RandomState(1).randint(0, 3, 1500)]
    exp_all, exp_cat = _oracle_means(X), _oracle_means(X, labels, ["t", "n"])
    for env in ({}, {"ICV_CHAIN_FAR": "700"}, {"ICV_CHAIN_FAR": "1"}, {"ICV_NO_CHAIN_QUEUES": "1"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        lib.icv_developer_knobs_reload()
        np.testing.assert_array_equal(_gpu_means(X), exp_all)
        np.testing.assert_array_equal(_gpu_means(X, labels, ["t", "n"], pieces=3), exp_cat)
        for k in env:
            monkeypatch.delenv(k)
    lib.icv_developer_knobs_reload()
    wide = sp.csr_matrix(_expr(300, 70001, seed=13, density=0.01))  # > 65 535 columns: 16-bit row offsets do not apply
    np.testing.assert_array_equal(_gpu_means(wide), _oracle_means(wide))


@pytest.mark.parametrize("n,g,density", [(64, 96, 0.9), (63, 96, 0.9), (65, 40, 0.5), (129, 31, 1.0), (640, 2000, 0.25),
                                         (10_000, 300, 0.07), (3000, 96, 0.001)])
def test_csr_queue_kernel_edges(n, g, density):
    """Round boundaries (64 rows), full columns (64 LDS rows a round), tiles narrower than a cache line, rows with more
    than 16 entries in a tile, rounds without any entry."""
    for dtype in (np.float32, np.float64):
        X = sp.csr_matrix(_expr(n, g, seed=n + g, dtype=dtype, density=density))
        np.testing.assert_array_equal(_gpu_means(X), _oracle_means(X))
    labels = np.array(["n", "t"])[np.random.RandomState(2).randint(0, 2, n)]
    X = sp.csr_matrix(_expr(n, g, seed=n + g, density=density))
    np.testing.assert_array_equal(_gpu_means(X, labels, ["t", "n"], pieces=2), _oracle_means(X, labels, ["t", "n"]))


@pytest.mark.parametrize("pieces", [2, 4, 8, 16])
def test_csr_queue_kernel_any_lanes_per_row_gives_scipys_bits(pieces, monkeypatch):
    """k_colchain_csrq<PIECES>: 8 / 16 / 32 / 64 entry slots per row and tile (rounds of 64 / 64 / 32 / 16 rows); the
    launcher picks from the density, ICV_CHAIN_PIECES forces one.  Every shape gives scipy's bits whatever the density --
    rows beyond the slots take the guarded loads -- including round edges of the short rounds and float64."""
    from infercnvpy_amd import _lib

    monkeypatch.setenv("ICV_CHAIN_PIECES", str(pieces))
    _lib.load().icv_developer_knobs_reload()
    for n, g, density, dtype in ((2000, 20000, 0.14, np.float32), (1000, 20000, 0.02, np.float32),
                                 (700, 3001, 0.4, np.float32), (97, 96, 1.0, np.float32), (33, 40, 0.6, np.float64),
                                 (1500, 9000, 0.21, np.float64), (17, 129, 0.9, np.float32)):
        X = sp.csr_matrix(_expr(n, g, seed=n + g + pieces, dtype=dtype, density=density))
        np.testing.assert_array_equal(_gpu_means(X), _oracle_means(X))
        labels = np.array(["n", "t"])[np.random.RandomState(2).randint(0, 2, n)]
        np.testing.assert_array_equal(_gpu_means(X, labels, ["t", "n"], pieces=2), _oracle_means(X, labels, ["t", "n"]))
    monkeypatch.delenv("ICV_CHAIN_PIECES")
    _lib.load().icv_developer_knobs_reload()


def test_csr_integer_counts_and_long_rows():
    rs = np.random.RandomState(3)
    Xi = rs.poisson(0.3, (900, 3000)).astype(np.int64)
    Xi[5] = rs.poisson(3.0, 3000)  # a row with (nearly) every column stored
    Xi[17] = 0                      # an empty row
    X = sp.csr_matrix(Xi)
    labels = np.array(["n", "t"])[rs.randint(0, 2, 900)]
    np.testing.assert_array_equal(_gpu_means(X), _oracle_means(X))
    np.testing.assert_array_equal(_gpu_means(X, labels, ["n", "t"], pieces=3), _oracle_means(X, labels, ["n", "t"]))


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64])
def test_csc_means_are_scipy_bit_for_bit(dtype):
    from infercnvpy_amd import _engine

    rs = np.random.RandomState(8)
    n, g = 3000, 1500
    D = _expr(n, g, seed=8, density=0.2)
    D[:, 7] = 0          # a column without entries
    D[:, 9] = rs.gamma(0.3, 1.0, n) + 0.1  # a full column: the pairwise tree at its largest
    D[:5, 11] = 1.0      # few entries
    D[5:, 11] = 0
    X = sp.csc_matrix(D.astype(dtype))
    np_dtype = np.float32 if dtype == np.float32 else np.float64
    got = _engine.csc_column_means(X, np_dtype=np_dtype, max_entries=200_000)
    np.testing.assert_array_equal(got, _oracle_means(X))
    labels = np.array(["n", "t", "u"])[rs.randint(0, 3, n)]
    groups = np.full(n, -1, np.int32)
    counts = []
    for gi, c in enumerate(["u", "n"]):
        groups[labels == c] = gi
        counts.append(int((labels == c).sum()))
    got = _engine.csc_column_means(X, groups, 2, counts, np_dtype=np_dtype)
    np.testing.assert_array_equal(got, _oracle_means(X, labels, ["u", "n"]))


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16385, 50_001])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_column_major_dense_mean_is_numpys_pairwise_order(n, dtype):
    """np.mean(X, axis=0) of an F-ordered matrix (reference :385 on the matrix as stored): numpy reduces every column with
    its contiguous inner loop -- pairwise summation over pieces of 8 192 elements -- not as the sequential chain a
    C-ordered matrix gets; icv_colsum_pairwise restates it."""
    from infercnvpy_amd import _engine

    g = 37
    X = np.asfortranarray(_expr(n, g, seed=n, dtype=dtype, density=0.6))
    got = _engine.fortran_column_means(X, max_bytes=max(1, n) * X.itemsize * 10)  # several column blocks
    exp = np.mean(X, axis=0)
    assert got.dtype == exp.dtype
    np.testing.assert_array_equal(got, exp)
    one = np.ascontiguousarray(X[:, :1])  # a single column: numpy reduces it as a 1-D contiguous array -- pairwise too
    np.testing.assert_array_equal(_engine.fortran_column_means(one), np.mean(one, axis=0))
    if n >= 129:  # the two layouts really are different sums
        Xs = np.asfortranarray(X[::2])  # and a row-sliced view keeps the layout rule (smaller stride innermost)
        np.testing.assert_array_equal(_engine.fortran_column_means(X[::2]), np.mean(X[::2], axis=0))
        np.testing.assert_array_equal(_engine.fortran_column_means(Xs), np.mean(Xs, axis=0))


def test_public_call_on_column_major_input_matches_numpy_order():
    """The whole call: an F-ordered adata.X (a transposed genes x cells array) gives the X_cnv of the oracle run on the
    same array (numpy's own mean), on one shard and on three; integer counts too; per-category means (X[rows, :] is
    C-ordered in numpy) unchanged."""
    import pandas as pd

    import cases
    import infercnvpy_amd as cnv
    from oracle import infercnv_oracle as O

    v = cases.synthetic_var([300, 120, 101, 60])
    n = 9000
    XT = np.ascontiguousarray(cases.synthetic_expr(n, len(v["names"]), seed=3).T)  # genes x cells, C-ordered
    X = XT.T  # cells x genes: column-major
    assert X.strides[0] < X.strides[1]
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    labels = np.array(["n", "t"])[np.random.RandomState(0).randint(0, 2, n)]
    obs = pd.DataFrame({"group": labels}, index=[f"c{i}" for i in range(n)])
    for Xin in (X, np.asfortranarray(np.random.RandomState(1).poisson(0.4, X.shape).astype(np.int64))):
        _, exp, _, _ = O.infercnv(Xin, v["chromosome"], v["start"], chunksize=3000)
        for kw in (dict(), dict(devices=[0, 0, 0])):
            _, got, _ = cnv.tl.infercnv(cnv.SimpleAnnData(Xin, obs=obs, var=var), inplace=False, chunksize=3000, **kw)
            np.testing.assert_array_equal(got.toarray() == 0, exp.toarray() == 0)
            np.testing.assert_allclose(got.toarray(), exp.toarray(), rtol=0, atol=1e-6)
    _, exp, _, _ = O.infercnv(X, v["chromosome"], v["start"], chunksize=3000, obs_col=labels, reference_cat=["n"])
    _, got, _ = cnv.tl.infercnv(cnv.SimpleAnnData(X, obs=obs, var=var), inplace=False, chunksize=3000,
                                reference_key="group", reference_cat=["n"])
    np.testing.assert_array_equal(got.toarray() == 0, exp.toarray() == 0)


@pytest.mark.parametrize("shape,kind", [((5000, 20000), "gamma"), ((3001, 1337), "gamma"), ((64, 96), "gamma"),
                                        ((1, 5), "gamma"), ((2500, 4099), "counts"), ((1500, 3000), "ties"),
                                        ((4000, 999), "scaled"), ((900, 2000), "negative"), ((700, 513), "nan")])
def test_chain_by_blocks_is_the_chain_bit_for_bit(shape, kind):
    """icv_colchain_blocks_* (integer block records + scan, csrc/icv_kernel_blocks.hpp) against the chain kernel and
    numpy: equal bits whatever the data -- integer counts (exact sums), values that tie in the upper binades, columns
    scaled over 60 decades, a negative entry, a NaN -- and whatever the estimate of the start (right, wrong binade, NaN)."""
    from infercnvpy_amd import _engine

    torch = _engine._torch()
    n, g = shape
    rs = np.random.RandomState(n + g)
    X = _expr(n, g, seed=n + g)
    if kind == "counts":
        X = np.floor(X * 4).astype(np.float32)
    elif kind == "ties":
        X = (np.round(X * 8) / 8 + (X > 0) * 1024).astype(np.float32)
    elif kind == "scaled":
        X = (X * np.float32(10.0) ** rs.randint(-30, 30, size=g).astype(np.float32)).astype(np.float32)
    elif kind == "negative":
        X[rs.randint(n), rs.randint(g)] = -1.5
    elif kind == "nan":
        X[n // 2, 7] = np.nan
    dm = _engine.to_device_matrix(X)
    want = _engine.column_chain(dm, None, None, n).cpu().numpy()
    with np.errstate(invalid="ignore"):
        ref = np.add.reduce(X, axis=0) if g > 1 and n >= 2 else want
    np.testing.assert_array_equal(want, ref)
    cb = _engine.ChainBlocks(dm)
    total = cb.sums()
    np.testing.assert_allclose(total.cpu().numpy(), X.sum(axis=0, dtype=np.float64), rtol=1e-12, atol=0)
    cb.records(None)
    got = cb.scan(torch.zeros(g, dtype=torch.float32, device="cuda")).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.int32), want.view(np.int32))
    # continued from another matrix's exact values, with estimates of every quality
    A = _expr(777, g, seed=3)
    sA = _engine.column_chain(_engine.to_device_matrix(A), None, None, 777)
    whole = _engine.column_chain(dm, sA.clone(), None, n).cpu().numpy()
    estA = torch.from_numpy(A.sum(axis=0, dtype=np.float64)).cuda()
    for est in (estA, estA * 2.1, estA * 0.3, torch.zeros_like(estA), torch.full_like(estA, float("nan"))):
        cb.records(est)
        got = cb.scan(sA.clone()).cpu().numpy()
        np.testing.assert_array_equal(got.view(np.int32), whole.view(np.int32))
    # column ranges (the ranks' pipeline) cover the columns
    cb.records(estA)
    acc = sA.clone()
    for c0, c1 in ((0, g // 3), (g // 3, g // 3), (g // 3, g)):
        cb.scan(acc, cols=(c0, c1))
    np.testing.assert_array_equal(acc.cpu().numpy().view(np.int32), whole.view(np.int32))


def test_chain_by_blocks_replays_little_at_config3_geometry():
    """125 000 x 20 000 (one rank's rows of config 3 at 8 ranks) continued from a running chain: the scan replays well
    under 1 % of the (block, column) pairs, and the result is the chain kernel's."""
    from infercnvpy_amd import _engine

    torch = _engine._torch()
    gen = torch.Generator(device="cuda").manual_seed(5)
    g = 20000

    def rows(n):  # (in 5000-row pieces, as bench.py: one 125 000 x 20 000 draw leaves the first ~16 000 rows zero)
        out = torch.empty((n, g), dtype=torch.float32, device="cuda")
        for r in range(0, n, 5000):
            x = torch._standard_gamma(torch.full((min(5000, n - r), g), 0.3, device="cuda"), generator=gen)
            out[r:r + 5000] = torch.where(x < 0.5, torch.zeros_like(x), x)
        return out

    A, B = rows(125_000), rows(125_000)
    dmA, dmB = _engine.DeviceMatrix(dense=A), _engine.DeviceMatrix(dense=B)
    sA = _engine.column_chain(dmA, None, None, 250_000)
    want = _engine.column_chain(dmB, sA.clone(), None, 250_000)
    cbA, cbB = _engine.ChainBlocks(dmA), _engine.ChainBlocks(dmB)
    tA = cbA.sums()
    cbA.records(None)
    gotA = cbA.scan(torch.zeros(g, dtype=torch.float32, device="cuda"))
    assert torch.equal(gotA.view(torch.int32), sA.view(torch.int32))
    cbB.sums()
    cbB.records(tA)
    gotB = cbB.scan(gotA.clone())
    assert torch.equal(gotB.view(torch.int32), want.view(torch.int32))
    assert int(cbB.replayed.item()) < 0.005 * cbB.n_blocks(), (int(cbB.replayed.item()), cbB.n_blocks())
    assert int(cbA.replayed.item()) < 0.03 * cbA.n_blocks(), (int(cbA.replayed.item()), cbA.n_blocks())

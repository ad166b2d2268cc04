"""CPU restatement of the arithmetic of the CSR stored-entries kernel (``k_smooth_se``, csrc/icv_kernel_se.hpp) from
the plan's host tables (``icv_plan_se_tables``): fixed-point block bins of the differences to the zero row,
per-wavefront prefix sums, the packed window words (csrc/icv_plan.hpp: ``se_window_words``) -- compared with the
oracle's smoothed windows (reference ``_running_mean_by_chromosome``, tl/_infercnv.py:301-356).  Runs without a GPU:
it pins the host side of the kernel (which LDS slots a window reads, which wavefront total it adds)."""
import numpy as np
import pytest

import cases
from infercnvpy_amd._plan import GenePlan
from oracle import infercnv_oracle as O

PLANE = 513
MAGIC = 6755399441055744.0  # 1.5 * 2^52


def _slot(b):
    return (b & 7) * PLANE + (b >> 3)


def _emulate(plan, window, ref, cols, vals, cap, k0, k1):
    """Windows (numerators / denominators applied) of ONE cell with stored entries (cols, vals) as the kernel forms them."""
    t = plan.se_tables()
    assert t is not None
    B, NB, W = int(plan.info.block), int(plan.info.n_blocks), plan.n_windows
    ref = ref.astype(np.float32)
    z = np.clip(np.float32(0) - ref, -np.float32(cap), np.float32(cap)).astype(np.float32)
    # ---- bins: integers, order independent
    S0 = np.zeros(NB, dtype=object)
    S1 = np.zeros(NB, dtype=object)
    for g, x in zip(cols, vals):
        b = int(t["col_block"][g])
        if b < 0:
            continue
        v = np.clip(np.float32(x) - ref[g], -np.float32(cap), np.float32(cap)).astype(np.float32)
        d = float(v) - float(z[g])
        j = int(t["col_offset"][g])
        S0[b] += int(np.rint(d * 2.0 ** k0))
        S1[b] += int(np.rint(d * (j * 2.0 ** k1)))
    r = 2.0 ** (k1 - k0)
    s0 = np.array([float(v) for v in S0])
    t1 = np.array([float(t["block_gene0"][b]) * r * float(S0[b]) + float(S1[b]) for b in range(NB)])
    # ---- prefix sums per wavefront of 512 blocks, totals
    slots = np.zeros((8 * PLANE + 1, 2))  # slot 512 of plane 0 stays zero
    tot = np.zeros((16, 2))
    for w in range((NB + 511) // 512):
        lo, hi = 512 * w, min(NB, 512 * (w + 1))
        p0, p1 = np.cumsum(s0[lo:hi]), np.cumsum(t1[lo:hi])
        for b in range(lo, hi):
            slots[_slot(b)] = (p0[b - lo], p1[b - lo])
        tot[w] = (p0[-1], p1[-1])
    # ---- zero-row window sums and the windows
    st, ln = plan.window_table()  # sorted-gene coordinates
    order = plan.order
    zs = z[order].astype(np.float64)
    out = np.zeros(W)
    for jw in range(W):
        w0, w1 = int(t["w0"][jw]), int(t["w1"][jw])
        pb = slots[w0 & 0x1FFF]
        pm = slots[(w0 >> 13) & 0x1FFF] + tot[(w0 >> 26) & 0xF]
        pe = slots[w1 & 0x1FFF] + tot[(w1 >> 27) & 0xF]
        sg = (w1 >> 13) & 0x3FFF
        flat = (w0 >> 31) & 1
        seg = zs[st[jw]: st[jw] + ln[jw]]
        if not flat:
            wts = O.window_weights(window).astype(np.float64)
            base = float(np.dot(wts, seg))
            a = (pm[1] - pb[1]) - (sg - 1) * r * (pm[0] - pb[0])
            d = (sg + window) * r * (pe[0] - pm[0]) - (pe[1] - pm[1])
            out[jw] = (base + (a + d) * 2.0 ** -k1) / wts.sum()
        else:
            out[jw] = (float(seg.sum()) + (pe[0] - pb[0]) * 2.0 ** -k0) / sg
    return out


def _fraction_bits(B, cap):
    import math

    k0 = 51 - math.frexp(B * 2.0 * cap + 1.0)[1]
    k1 = 51 - math.frexp(B * (B - 1) * cap + 1.0)[1]
    return k0, k1


@pytest.mark.parametrize("genes,window,step,extra", [
    (cases.GENES_PER_CHROM_20K, 250, 10, (("chrX", 31), (None, 3))),   # config 4: windows cross wavefront ranges
    (cases.GENES_PER_CHROM_20K, 100, 10, ()),
    ([1500, 700, 333, 90], 120, 4, ((None, 5),)),                      # B = 4, a flat window
    ([600, 260, 251, 250, 249, 90], 100, 2, (("chrM", 3),)),           # B = 2: S1 finer than S0
])
def test_stored_entries_arithmetic_matches_the_oracle_windows(genes, window, step, extra):
    v = cases.synthetic_var(genes, extra=extra)
    n_genes = len(v["names"])
    plan = GenePlan(v["chromosome"], v["start"], window_size=window, step=step, exclude_chromosomes=None)
    B = int(plan.info.block)
    assert B > 1 and plan.se_tables() is not None
    rng = np.random.RandomState(5)
    cap = 3.0
    k0, k1 = _fraction_bits(B, cap)
    ref = rng.gamma(0.3, 0.2, size=n_genes).astype(np.float32)
    for nnz in (0, 1, 40, n_genes // 12):
        cols = np.sort(rng.choice(n_genes, size=nnz, replace=False))
        vals = (rng.gamma(0.5, 1.5, size=nnz) + 0.01).astype(np.float32)
        x = np.zeros((1, n_genes), dtype=np.float32)
        x[0, cols] = vals
        got = _emulate(plan, window, ref, cols, vals, cap, k0, k1)
        centred = np.clip(x - ref[None, :], -np.float32(cap), np.float32(cap))
        _, exp = O.smooth_all_chromosomes(centred, v["chromosome"], v["start"], window, step)
        assert got.shape == exp[0].shape
        np.testing.assert_allclose(got, exp[0], rtol=0, atol=2e-12)


def test_geometries_without_the_stored_entries_kernel():
    v = cases.synthetic_var([300, 120, 101])
    assert GenePlan(v["chromosome"], v["start"], window_size=101, step=7).se_tables() is None   # odd window: B = 1
    assert GenePlan(v["chromosome"], v["start"], window_size=100, step=1).se_tables() is None   # gcd(1, 50) = 1
    assert GenePlan(v["chromosome"], v["start"], window_size=100, step=10).se_tables() is not None

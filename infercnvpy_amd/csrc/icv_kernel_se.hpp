// k_smooth_se: CSR float32 input in block form -- the work of a cell is proportional to its STORED ENTRIES.
// (Second generation of the round-2 kernel k_smooth_sd, which it replaces.)
//
// A cell without stored entries is the pre-centred zero row z = clip(0 - ref), whose window sums
// ("base") are the same for every cell; a stored entry changes gene g from z[g] to v = clip(x - ref[g]), windows are
// linear in the gene values, so only the differences d = v - z[g] of the stored entries are accumulated -- into
// per-block bins {S0 = sum d, S1 = sum j d} (j: gene offset inside the block) in 64-bit FIXED POINT with LDS integer
// atomics (order-independent, bit-reproducible) -- the bins become float64 prefix sums over the blocks, and a pyramid
// window is a linear combination of the prefix sums at three blocks.
//
// What changed against k_smooth_sd (PMC counters: 6 050 VALU + 1 890 SALU wave-instructions per cell there, 4 060 +
// 1 500 here):
//   * prefix sums are kept PER WAVEFRONT (thread t owns blocks 8 t .. 8 t + 7); a window adds the total of one
//     wavefront where it crosses into the next (plan table, icv_plan.hpp: se_window_words).  No second-level scan,
//     one barrier less (four per cell), smaller magnitudes in the differences;
//   * the three LDS slots of a window, its gene offset and the zero-row sum come in ONE 16-byte table entry per
//     window (k_se_wtab), the first-gene offsets of a thread's blocks (float64, times r) in four 16-byte loads;
//   * the bins start every cell at the bit pattern of 1.5 * 2^52 and stay below 2^51 in magnitude (S0 in units of
//     2^-k0, S1 of 2^-k1, k0 / k1 from the clip value and the block size: se_fraction_bits): the scan turns a bin
//     into float64 with ONE subtraction instead of a 64-bit integer conversion;
//   * the entry's contribution to S1 is fma(d, j 2^k1, magic) with j 2^k1 a float of the column table -- no 64-bit
//     integer multiply; the block's first-gene offset enters scaled by r = 2^(k1-k0);
//   * median: 4096-bin fine histogram (16-bit) + 64 coarse bins; ONE wavefront resolves both middle ranks with two
//     DPP prefix sums (as k_smooth_x16), the others only gather their candidate windows; one wavefront ranks them;
//   * noise-threshold moments are accumulated per thread over the cells of a chunk (CHUNK) -- no per-cell wavefront
//     reductions, no per-cell moment traffic;
//   * no per-phase priorities, tables and flags read once per phase.
//
// Per workgroup (512 threads, two workgroups per CU) iteration `it` handles three cells at once; every single-wavefront
// step of the median sits beside bulk work of all wavefronts (what bounds the kernel, and everything that was measured
// not to: DESIGN.md 4.4, profiles/r03_se_late_experiments.txt, r03_se_occupancy.txt):
//   phase 0   cell it: bins reset to the bias pattern     | cell it-2: x_res, moments | wavefront 0: coarse bins of the
//   barrier A                                                                           two middle ranks of cell it-1
//   phase 1   cell it: stored entries -> bins (2 ds_add_u64 per entry)                | wavefront 0: their fine bins
//   barrier B1
//   phase 2   cell it: bins -> float64 prefix sums per wavefront; window table,       | cell it-1: windows in the
//             {column, value} of cell it+1 requested                                    middle bins gathered (<= 64)
//   barrier B2
//   phase 3   cell it: windows, histogram atomics; table entries of cell it+1         | last wavefront: exact ranks of
//   barrier B3         gathered                                                         the candidates -> median of it-1
// A NaN among the stored values of a cell (never on real data) and a cell with more than 64 windows in its median
// bins are handed back to the generic k_smooth (row_list).
#pragma once
#include <cstddef>

#include "icv_kernel_ws.hpp"
#include "icv_kernel_x16.hpp"
#include "icv_plan.hpp"

namespace icv {

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_shift0(double v) {  // lanes without a source receive 0
    // all rows written: bound_ctrl supplies the zeros (no v_mov of the old value); masked rows need old = 0
    constexpr bool kBound = ROWMASK == 0xf;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, kBound);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, kBound);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sums over the 64 lanes in float64, fixed order (row shifts 1, 2, 4, 8, then the row totals)
__device__ __forceinline__ double wave_scan_f64(double v) {
    v += dpp_shift0<0x111, 0xf>(v);
    v += dpp_shift0<0x112, 0xf>(v);
    v += dpp_shift0<0x114, 0xf>(v);
    v += dpp_shift0<0x118, 0xf>(v);
    v += dpp_shift0<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
    v += dpp_shift0<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
    return v;
}
// The bins {S0,S1} of block b, and then the prefix sums up to b, live at slot b >> 3 of plane b & 7 (8 planes of 512
// 16-byte pairs, plane stride 513 pairs = icv_plan.hpp: se_slot): the eight accesses of a thread that owns blocks
// 8t .. 8t+7 (one plane, consecutive slots over the lanes) are conflict-free
constexpr int kWsPlane = kSePlane;
constexpr int kSdPlaneBytes = 16 * 8 * kWsPlane;

// PF (template parameter, 1 .. 4): stored entries prefetched per thread = PF x 512 per cell; longer rows fetch the rest
// inside phase 1.  The launcher picks it from the mean row length (+ 4 sigma of a binomial): every slot costs ~22 VALU
// instructions per thread and cell whether it holds an entry or not, so sparser matrices run fewer of them.
constexpr int kSeCoarse = NBIN / 64, kSeRep = 4;     // coarse histogram: 64 bins x 4 replicas (lane & 3)
// LDS map: scratch (the wavefront totals first: offset 0) | coarse histogram | planes | fine histogram.  Everything but
// the fine histogram lies below 64 KB, so that its base travels in the offset field of the LDS instruction
constexpr int kSeScratchOff = 0;
constexpr int kSeScratchBytes = 1024;
constexpr int kSeCoarseOff = kSeScratchOff + kSeScratchBytes;
constexpr int kSePlanesOff = kSeCoarseOff + kSeCoarse * kSeRep * 4;
constexpr int kSeHistOff = kSePlanesOff + kSdPlaneBytes;  // fine histogram: NBIN 16-bit counters
constexpr int kSeLds = kSeHistOff + NBIN * 2;
static_assert(kSePlanesOff + 7 * 16 * kWsPlane < 65536, "plane offsets must fit the LDS offset field");

struct ScratchE {
    double2 tot[16];  // {S0, T1} totals of the wavefronts' blocks; entries 8 .. 15 stay zero (se_window_words)
    double cand[64];
    double med[2];    // the two middle order statistics of the previous cell
    int sel[8];       // b1, b2, windows below b1, windows in b1, windows in b2, state (0 rank, 1 NaN, 2 handed back)
    int ncand;
    int nanflag;      // a window of the current cell is NaN
    int bad[2];       // a stored NaN in the cell of an even / odd iteration
    int st_out;       // state of the cell whose median sits in med[] (phase 0 of the next iteration reads both)
    unsigned long long tacc[9];  // -DICV_SE_PROFILE: shader cycles per phase / barrier wait of one wavefront
};
static_assert(sizeof(ScratchE) <= kSeScratchBytes, "ScratchE must fit the scratch region");

// per input column {LDS byte address of the block's bins (all ones: masked column), ref_lo, j 2^k1 as float,
// z = clip(0 - ref)}; bounded references: ref_hi in a second table
__global__ void __launch_bounds__(256) k_se_table(const KParams P, u32x4* tab, float* tab_hi, float scale1) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P.n_cols) return;
    const float* lo = static_cast<const float*>(P.ref_lo);
    const float* hi = P.bounded ? static_cast<const float*>(P.ref_hi) : lo;
    const int pos = P.dst[g];
    u32x4 e = {0xffffffffu, 0u, 0u, 0u};
    if (pos >= 0) {
        const int blk = pos / P.B;
        e.x = (uint32_t)(kSePlanesOff + 16 * se_slot(blk));
        e.y = __float_as_uint(lo[g]);
        e.z = __float_as_uint((float)(pos - blk * P.B) * scale1);
        e.w = __float_as_uint(centre_clip<float>(0.0f, lo[g], hi[g], (float)P.cap, P.bounded, P.trunc));
    }
    tab[g] = e;
    if (tab_hi) tab_hi[g] = hi[g];
}

// per window {w0, w1 (plan: se_window_words), zero-row window sum as float64}: the numerator before the division by
// the pyramid weight sum / the gene count, canonical order
__global__ void __launch_bounds__(256) k_se_wtab(const KParams P, const float* zrow, const uint32_t* w0,
                                                 const uint32_t* w1, u32x4* wt, double* g_r, double r) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < P.NB + 8) g_r[j] = j < P.NB ? (double)P.blk_g0[j] * r : 0.0;  // first-gene offset of block j, times 2^(k1-k0)
    if (j >= P.W) return;
    const int wp = P.w_pack[j];
    const int ln = wp >> 16;
    const float* z = zrow + (wp & 0xffff) * P.B;
    double acc = 0.0;
    if (ln > 0) {
        const int h = ln / 2;
        for (int k = 0; k < ln; ++k) acc = fma((double)(k < h ? k + 1 : ln - k), (double)z[k], acc);
    } else {
        for (int k = 0; k < -ln; ++k) acc = acc + (double)z[k];
    }
    const u32x4 e = {w0[j], w1[j], (uint32_t)__double2loint(acc), (uint32_t)__double2hiint(acc)};
    wt[j] = e;
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f64x2 lds_f64x2_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
// 16 bytes at an absolute LDS byte offset (the dynamic LDS of the kernel starts at 0)
#define ICV_LDS_D2_AT(OFF) (*reinterpret_cast<const lds_f64x2_t*>(static_cast<uintptr_t>(OFF)))
typedef __attribute__((address_space(3))) unsigned lds_u32_t;
#define ICV_LDS_ADD_U32(OFF, V)                                                                                   \
    __hip_atomic_fetch_add(reinterpret_cast<lds_u32_t*>(static_cast<uintptr_t>(OFF)), (V), __ATOMIC_RELAXED, \
                           __HIP_MEMORY_SCOPE_WORKGROUP)

// WIN: the float64 windows of every cell (before centring) also go to P.win_out[cell * P.win_ld + j]
// (calculate_gene_values, reference tl/_infercnv.py:274-288); cells handed back are rewritten by k_smooth
template <int MAXW, bool CHUNK, bool BOUNDED, int PF = 4, bool WIN = false>
__global__ void __launch_bounds__(NT, 4) k_smooth_se(const KParams P) {
    constexpr int kSePF = PF;
    static_assert(PF >= 1 && PF <= 4, "entry slots per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2* SP = reinterpret_cast<double2*>(smem + kSePlanesOff);
    unsigned* hist = reinterpret_cast<unsigned*>(smem + kSeHistOff);      // two 16-bit bins per word
    unsigned* coarse = reinterpret_cast<unsigned*>(smem + kSeCoarseOff);  // [bin][replica]
    ScratchE* sc = reinterpret_cast<ScratchE*>(smem + kSeScratchOff);

    const int t = threadIdx.x;
    const int W = P.W, NB = P.NB;
    const int k1 = (W - 1) / 2, k2 = W / 2;
    const float inv_bound = (float)(1.0 / P.med_bound);
    const double pyr_den = P.pyr_den, pyr_rcp = P.pyr_rcp;
    const double q0inv = P.sd_qinv, q1inv = P.sd_q1inv;  // 2^-k0, 2^-k1
    const double scale0 = P.sd_scale;
    const double rr = P.sd_r;                                // 2^(k1-k0)
    const double win_r = (double)(P.sd_window + 1) * rr;     // (s + n) r = (s - 1) r + (n + 1) r
    const float cap = (float)P.cap;
    const int64_t n_mine = (P.n_rows - blockIdx.x + gridDim.x - 1) / gridDim.x;

    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(smem)) != 0)
        __builtin_trap();  // bins and prefix sums are addressed by absolute LDS offsets (column / window tables)

    // everything zero once: bins incl. the never-written slot 512 of every plane, histograms, scratch
    for (int i = t; i < kSeLds / 16; i += NT) reinterpret_cast<int4*>(smem)[i] = make_int4(0, 0, 0, 0);
    // this wavefront owns blocks at all (uniform): geometries with few blocks skip the unused tail of the planes
    const bool wblk = ((t & ~63) * 8) < NB;

    const __amdgpu_buffer_rsrc_t tab_rs = make_rsrc(P.sd_tab, (unsigned)P.n_cols * 16u);
    const __amdgpu_buffer_rsrc_t thi_rs = make_rsrc(BOUNDED ? P.sd_tab_hi : P.sd_tab, BOUNDED ? (unsigned)P.n_cols * 4u : 0u);
    const __amdgpu_buffer_rsrc_t wt_rs = make_rsrc(P.sd_wtab, (unsigned)W * 16u);
    const __amdgpu_buffer_rsrc_t g16_rs = make_rsrc(P.sd_g16, (unsigned)(NB + 8) * 8u);
    // row offsets travel through vector loads (a provably uniform address becomes a scalar load, whose counter is
    // shared with the LDS operations of the phase that follows)
    int zoff = 0;
    asm volatile("" : "+v"(zoff));

    // Software pipeline over the cells of this workgroup; iteration `it`:
    //   phase 0   bins of cell it zeroed | x_res of cell it-2 | wavefront 0: coarse bins of the middle ranks of it-1
    //   phase 1   entries of cell it     | wavefront 0: fine bins of the middle ranks of it-1
    //   phase 2   prefix sums of cell it | candidates of it-1 gathered, histograms cleared; loads for it+1
    //   phase 3   windows of cell it     | last wavefront: exact ranks of the candidates of it-1 -> its median
    // Every single-wavefront step of the median sits beside bulk work of all wavefronts; the windows of a cell stay in
    // registers for two iterations (two register sets, the loop is unrolled by two).
    int64_t a0 = 0;  // first stored entry of the cell of phase 1, and its entry count
    int c0 = 0;
    unsigned nidx[kSePF];
    float nval[kSePF];
    u32x4 etab[kSePF];  // table entries of the columns of the cell of the NEXT phase 1 (gathered a phase 3 ahead)
    float ehi[kSePF];
#pragma unroll
    for (int i = 0; i < kSePF; ++i) {
        nidx[i] = 0u;
        nval[i] = 0.0f;
        etab[i] = u32x4{0xffffffffu, 0u, 0u, 0u};
        ehi[i] = 0.0f;
    }
    // windows of the cells of even / odd iterations, and their histogram bins (two 16-bit bins per register, 0xffff:
    // no window)
    double wvE[MAXW], wvO[MAXW];
    unsigned wbE[(MAXW + 1) / 2], wbO[(MAXW + 1) / 2];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) wvE[i] = wvO[i] = 0.0;
#pragma unroll
    for (int i = 0; i < (MAXW + 1) / 2; ++i) wbE[i] = wbO[i] = 0xffffffffu;
    // CHUNK: running moments of this thread's windows over the cells of the current chunk
    double accS = 0.0, accQ = 0.0;
    int64_t chunk_cur = -1, chunk_end = INT64_MIN;
    (void)accS;
    (void)accQ;
    (void)chunk_cur;
    (void)chunk_end;
    // phase timers (developer build, -DICV_SE_PROFILE=<thread>): one thread accumulates s_memtime deltas in LDS
#ifdef ICV_SE_PROFILE
    if (t == ICV_SE_PROFILE)
        for (int i = 0; i < 9; ++i) sc->tacc[i] = 0;
#define ICV_SEP(i)                                              \
    if (P.dbg && t == ICV_SE_PROFILE) {                         \
        unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        sc->tacc[i] += now_ - sc->tacc[8];                      \
        sc->tacc[8] = now_;                                     \
    }
#else
#define ICV_SEP(i)
#endif
    // (Static wave priority by age -- the arbiter prefers the older wavefronts of a SIMD, so wavefront 7 reaches every
    // barrier last -- was measured in three gradings: it moves the barrier waits to other wavefronts and leaves the
    // kernel time unchanged, 6.3 ms: the SIMDs' instruction issue is what bounds the kernel.)
    __syncthreads();
#ifdef ICV_SE_PROFILE
    if (P.dbg && t == ICV_SE_PROFILE) sc->tacc[8] = __builtin_amdgcn_s_memtime();
#endif

    // wvA / wbA: registers of the cells with the parity of `it` (cell it-2 leaves them in phase 0, cell it enters in
    // phase 3); wvB / wbB: cell it-1
    auto iteration = [&](int64_t it, double (&wvA)[MAXW], unsigned (&wbA)[(MAXW + 1) / 2], double (&wvB)[MAXW],
                         unsigned (&wbB)[(MAXW + 1) / 2]) __attribute__((always_inline)) {
        const bool more = it >= 0 && it < n_mine;           // cell it: entries, prefix sums, windows
        const bool have1 = it >= 1 && it - 1 < n_mine;      // cell it-1: middle bins, candidates, ranks
        const bool have2 = it >= 2 && it - 2 < n_mine;      // cell it-2: x_res
        const int64_t cell = (int64_t)blockIdx.x + it * gridDim.x;
        const int par = (int)(it & 1);
        int tl = t;
        asm volatile("" : "+v"(tl));  // keep thread-derived values out of LICM (register budget)

        // ---------------- phase 0: x_res (cell it-2); coarse bins (cell it-1); bins zeroed (cell it) -------------
        if (more && wblk) {  // both bins of a block start at the bit pattern of 1.5 * 2^52
#pragma unroll
            for (int k = 0; k < 8; ++k)
                reinterpret_cast<int4*>(SP)[k * kWsPlane + tl] = make_int4(0, 0x43380000, 0, 0x43380000);
        }
        // wavefront 0, first half of the median search of cell it-1: the coarse bins of its two middle ranks
        int ch_st = 0, ch_C0 = 0, ch_C1 = 0, ch_b0 = 0, ch_b1 = 0;  // state, coarse bins, windows below them
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_SE_EXP_NOMEDIAN)  // upper bound: no median search at all (wrong results)
        if (false) {
#else
        if (have1 && tl < 64) {
#endif
            const int nanf = sc->nanflag, badf = sc->bad[par ^ 1];
            const uint4 c4 = reinterpret_cast<const uint4*>(coarse)[tl];
            if (badf) {
                ch_st = 2;
            } else if (nanf) {
                ch_st = 1;
            } else {
                const int c = (int)((c4.x + c4.y) + (c4.z + c4.w));
                const int cincl = wave_scan_dpp(c);
                ch_C0 = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(cincl > k1));
                ch_C1 = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(cincl > k2));
                ch_b0 = __builtin_amdgcn_readlane(cincl - c, ch_C0);
                ch_b1 = __builtin_amdgcn_readlane(cincl - c, ch_C1);
            }
            if (tl == 0) {
                sc->nanflag = 0;
                sc->bad[par ^ 1] = 0;
            }
        }
        if (have2) {
            const int64_t ocell = cell - 2 * (int64_t)gridDim.x;
            const double2 mm = *reinterpret_cast<const double2*>(sc->med);
            const int st2 = sc->st_out;
            const double med = (k1 == k2) ? mm.x : (mm.x + mm.y) / 2.0;
            if constexpr (CHUNK) {
                if (ocell >= chunk_end) {  // uniform: first cell of this workgroup in a new chunk
                    if (chunk_cur >= 0) {
                        const double2 m2 = wave_moments(accS, accQ);
                        if ((tl & 63) == 0)
                            reinterpret_cast<double2*>(P.chunk_part)[(chunk_cur * gridDim.x + blockIdx.x) * NWAVE + (tl >> 6)] = m2;
                    }
                    chunk_cur = (ocell + P.row_phase) / P.chunksize;
                    chunk_end = (chunk_cur + 1) * P.chunksize - P.row_phase;
                    accS = 0.0;
                    accQ = 0.0;
                }
            }
            // (a cell that was handed back gets med = 0 here and is rewritten by k_smooth afterwards)
            const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(P.out + ocell * P.ldo, (unsigned)W * 4u);
            double sum = 0.0, sq = 0.0;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                if (tl + i * NT < W) {  // (the store is range-checked by its descriptor as well)
                    const double y = wvA[i] - med;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)y), o_rs, (unsigned)tl * 4u, i * NT * 4, 0);
                    if constexpr (WIN) P.win_out[ocell * P.win_ld + tl + i * NT] = wvA[i];
                    sum = sum + y;
                    sq = fma(y, y, sq);
                }
            }
            if constexpr (CHUNK) {
                if (st2 != 2) {  // a cell handed back to k_smooth is accounted there
                    accS = accS + sum;
                    accQ = accQ + sq;
                }
            } else {
                const double2 m2 = wave_moments(sum, sq);
                if ((tl & 63) == 0) reinterpret_cast<double2*>(P.cell_part)[ocell * NWAVE + (tl >> 6)] = m2;
            }
            if (tl == 0) P.cell_median[ocell] = med;
        }
        ICV_SEP(0)
        __syncthreads();  // A: bins zero; x_res of cell it-2 out of the registers; coarse histogram consumed
        ICV_SEP(1)
        asm volatile("" : "+v"(tl));

        // ---------------- phase 1: fine bins (cell it-1); entries added (cell it) ---------------------------------
        // row offsets of cell it+1 (consumed in phase 2; unconditional: row 0 when there is no such cell)
        const int64_t nxt = cell + gridDim.x;
        const bool has_next = nxt >= 0 && nxt < P.n_rows;
        const int64_t nrow = has_next ? nxt : 0;
        const int64_t n0 = P.indptr[nrow + zoff], n1 = P.indptr[nrow + 1 + zoff];
        // first-gene offsets of the thread's blocks (times r, float64), consumed in phase 2
        u32x4 g16q[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if (more) {
#pragma unroll
            for (int k = 0; k < 4; ++k) g16q[k] = __builtin_amdgcn_raw_buffer_load_b128(g16_rs, (unsigned)tl * 64u, k * 16, 0);
        }
        if (have1 && tl < 64) {
            // ---- wavefront 0, second half: the fine bins inside the located coarse bins
            int4 r = make_int4(0, 0, 0, 0);
            int c2n = 0, st = ch_st;
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_SE_EXP_NOMEDIAN)
            if (false) {
#else
            if (st == 0) {
#endif
                int Cprev = -1, f = 0, fincl = 0;
                int res[2][3];
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const int k = which == 0 ? k1 : k2;
                    const int C = which == 0 ? ch_C0 : ch_C1;
                    const int belowC = which == 0 ? ch_b0 : ch_b1;
                    if (C != Cprev) {  // wavefront-uniform
                        const unsigned word = hist[(C * 64 + tl) >> 1];
                        f = (int)((word >> ((tl & 1) * 16)) & 0xffffu);
                        fincl = wave_scan_dpp(f);
                        Cprev = C;
                    }
                    const int j = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(fincl + belowC > k));
                    res[which][0] = C * 64 + j;
                    res[which][1] = __builtin_amdgcn_readlane(fincl - f, j) + belowC;
                    res[which][2] = __builtin_amdgcn_readlane(f, j);
                }
                r = make_int4(res[0][0], res[1][0], res[0][1], res[0][2]);
                c2n = res[1][2];
                // more than 64 windows share the median bins: the generic kernel recomputes the cell
                if (res[0][2] + (res[1][0] != res[0][0] ? res[1][2] : 0) > 64) st = 2;
            }
            if (tl == 0) {
                *reinterpret_cast<int4*>(sc->sel) = r;
                *reinterpret_cast<int2*>(sc->sel + 4) = make_int2(c2n, st);
                sc->ncand = 0;
                if (st == 2) {
                    const int slot = atomicAdd(P.row_count, 1);
                    P.row_list[slot] = cell - (int64_t)gridDim.x;
                }
            }
        }
        if (more) {
            int bad = 0;
            const auto add_entry = [&](const u32x4 e, float hi, float x) {
                if (e.x != 0xffffffffu) {  // masked columns contribute nothing
                    const float lo = __uint_as_float(e.y);
                    const float v = centre_clip<float>(x, lo, BOUNDED ? hi : lo, cap, BOUNDED ? 1 : 0, P.trunc);
                    const double d = (double)v - (double)__uint_as_float(e.w);
                    if (d != d) {
                        bad = 1;
                    } else {
                        // round to nearest integer, |.| < 2^51: the low mantissa bits of x + 1.5 * 2^52
                        const long long q0 = __double_as_longlong(fma(d, scale0, 6755399441055744.0)) - 0x4338000000000000ll;
                        const long long q1 = __double_as_longlong(fma(d, (double)__uint_as_float(e.z), 6755399441055744.0)) -
                                             0x4338000000000000ll;
                        lds_u64_t* bin = reinterpret_cast<lds_u64_t*>(static_cast<uintptr_t>(e.x));
                        __hip_atomic_fetch_add(bin, (unsigned long long)q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(bin + 1, (unsigned long long)q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < kSePF; ++i)
                if (tl + i * NT < c0) add_entry(etab[i], ehi[i], nval[i]);
            for (int k = tl + kSePF * NT; k < c0; k += NT) {  // rows with more than 2048 stored entries
                const int g = P.indices[a0 + k];
                add_entry(reinterpret_cast<const u32x4*>(P.sd_tab)[g], BOUNDED ? P.sd_tab_hi[g] : 0.0f,
                          static_cast<const float*>(P.values)[a0 + k]);
            }
            if (bad) sc->bad[par] = 1;  // benign race: every writer stores 1
        }
        ICV_SEP(2)
        __syncthreads();  // B1: bins complete; middle bins of cell it-1 published, fine histogram consumed
        ICV_SEP(3)
        asm volatile("" : "+v"(tl));

        // ---------------- phase 2: candidates (cell it-1); block prefix sums (cell it); loads for cell it+1 ----------
        // loads first: the window table of phase 3, then the entries of the next cell; all unconditional buffer loads
        // (empty range where there is nothing to do): exact counter bookkeeping
        u32x4 wt[MAXW];
        {
#pragma unroll
            for (int i = 0; i < MAXW; ++i)  // out-of-range windows read 0
                wt[i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rs, (unsigned)tl * 16u, i * NT * 16, 0);
            const int64_t u0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(n0 >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)n0);
            const int64_t u1 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(n1 >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)n1);
            const int64_t cnt = has_next ? u1 - u0 : 0;
            const unsigned npf = (unsigned)(cnt < kSePF * NT ? cnt : kSePF * NT);
            const __amdgpu_buffer_rsrc_t i_rs = make_rsrc(P.indices + u0, npf * 4u);
            const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(static_cast<const float*>(P.values) + u0, npf * 4u);
#pragma unroll
            for (int i = 0; i < kSePF; ++i) {
                nidx[i] = __builtin_amdgcn_raw_buffer_load_b32(i_rs, (unsigned)tl * 4u, i * NT * 4, 0);
                nval[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(v_rs, (unsigned)tl * 4u, i * NT * 4, 0));
            }
            a0 = u0;
            c0 = (int)(cnt < 0x7fffffff ? cnt : 0x7fffffff);
        }
        int st1 = 0;
        if (have1) {
            const int4 s = *reinterpret_cast<const int4*>(sc->sel);       // b1, b2, below, c1
            const int2 s2 = *reinterpret_cast<const int2*>(sc->sel + 4);  // c2, state
            st1 = s2.y;
            if (s2.y == 0) {
                // no window lies in a bin strictly between the bins of two adjacent ranks: b1 <= bin <= b2 selects them
                const unsigned span = (unsigned)(s.y - s.x);
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    const unsigned b = (wbB[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
                    if (b - (unsigned)s.x <= span) {  // 0xffff (no window) never passes
                        const int idx = atomicAdd(&sc->ncand, 1);
                        if (idx < 64) sc->cand[idx] = wvB[i];
                    }
                }
            }
            // both histograms of cell it-1 are consumed: clear them for the windows of cell it (phase 3)
            reinterpret_cast<int4*>(hist)[tl] = make_int4(0, 0, 0, 0);
            if (tl < kSeCoarse * kSeRep / 4) reinterpret_cast<int4*>(coarse)[tl] = make_int4(0, 0, 0, 0);
        }
        if (more && wblk) {
            // {S0 [2^-k0], T1 = g0 r S0 + S1 [2^-k1]} of the thread's blocks 8 t .. 8 t + 7; their prefix sums
            // over the blocks of this WAVEFRONT go back in place, the wavefront's totals to the scratch
            double s0[8], s1[8], t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double2 v = SP[k * kWsPlane + tl];  // 1.5 * 2^52 + the bin's integer, exactly
                s0[k] = v.x - 6755399441055744.0;
                const double g0r = (k & 1) ? __hiloint2double((int)g16q[k >> 1].w, (int)g16q[k >> 1].z)
                                           : __hiloint2double((int)g16q[k >> 1].y, (int)g16q[k >> 1].x);
                s1[k] = fma(g0r, s0[k], v.y - 6755399441055744.0);
                t0 = t0 + s0[k];
                t1 = t1 + s1[k];
            }
            const double y0 = wave_scan_f64(t0), y1 = wave_scan_f64(t1);
            double r0 = y0 - t0, r1 = y1 - t1;  // the lanes before this one
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                r0 = r0 + s0[k];
                r1 = r1 + s1[k];
                SP[k * kWsPlane + tl] = make_double2(r0, r1);
            }
            if ((tl & 63) == 63) sc->tot[tl >> 6] = make_double2(y0, y1);
        }
        ICV_SEP(4)
        __syncthreads();  // B2: prefix sums and wavefront totals complete; candidates of cell it-1 complete
        ICV_SEP(5)
        asm volatile("" : "+v"(tl));

        // ---------------- phase 3: ranks -> median (cell it-1); windows + histogram (cell it) ------------------------
        // the table entries of the columns of cell it+1 (its {column, value} were requested in phase 2): one L1 line
        // per entry, in flight behind the windows, consumed in the next phase 1
#pragma unroll
        for (int i = 0; i < kSePF; ++i) {
            etab[i] = __builtin_amdgcn_raw_buffer_load_b128(tab_rs, nidx[i] * 16u, 0, 0);
            ehi[i] = BOUNDED ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(thi_rs, nidx[i] * 4u, 0, 0)) : 0.0f;
        }
        if (have1 && (tl >> 6) == NWAVE - 1) {
            // ---- last wavefront (the one with the fewest windows): exact float64 ranks of the <= 64 candidates
            const int lane = tl & 63;
            const int4 s = *reinterpret_cast<const int4*>(sc->sel);
            const int ncr = sc->ncand;
            const double mine_raw = sc->cand[lane];
            double ma = 0.0, mb = 0.0;  // handed-back cell: placeholder, rewritten by k_smooth
            if (st1 == 1) {
                ma = mb = __builtin_nan("");
#if defined(ICV_DEV_EXPERIMENTS) && (defined(ICV_SE_EXP_NOMEDIAN) || defined(ICV_SE_EXP_NORANK))
            } else if (false) {  // upper bound: the ranking step costs nothing (wrong results)
#else
            } else if (st1 == 0) {
#endif
                const int n = ncr < 64 ? ncr : 64;
                const double mine = (lane < n) ? mine_raw : __builtin_inf();
                int r = 0;
                // n is wavefront-uniform (5 .. 20 windows share the median bins); the other candidates come as LDS
                // broadcast reads, four in flight (lane reads through SGPRs serialise: measured 1 700 cycles here)
                int q = 0;
                for (; q + 4 <= n; q += 4) {
                    const double2 oa = *reinterpret_cast<const double2*>(sc->cand + q);
                    const double2 ob = *reinterpret_cast<const double2*>(sc->cand + q + 2);
                    r += (int)(oa.x < mine) | ((int)(oa.x == mine) & (int)(q < lane));
                    r += (int)(oa.y < mine) | ((int)(oa.y == mine) & (int)(q + 1 < lane));
                    r += (int)(ob.x < mine) | ((int)(ob.x == mine) & (int)(q + 2 < lane));
                    r += (int)(ob.y < mine) | ((int)(ob.y == mine) & (int)(q + 3 < lane));
                }
                for (; q < n; ++q) {
                    const double o = sc->cand[q];
                    r += (int)(o < mine) | ((int)(o == mine) & (int)(q < lane));
                }
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(lane < n && r == k1 - s.z);
                const unsigned long long m2 = __builtin_amdgcn_ballot_w64(lane < n && r == k2 - s.z);
                ma = readlane_d(mine, m1 ? (int)__builtin_ctzll(m1) : 0);
                mb = readlane_d(mine, m2 ? (int)__builtin_ctzll(m2) : 0);
            }
            if (lane == 0) {
                *reinterpret_cast<double2*>(sc->med) = make_double2(ma, mb);
                sc->st_out = st1;
            }
        }
        if (more) {
            // a pyramid window of n genes starting at gene s of its chromosome, with b- the block before it, m1 / m2
            // the last block of its first / second half and {P0, P1} the prefix sums of {S0, T1}:
            //   [P1(m1) - P1(b-) - (s - 1)(P0(m1) - P0(b-))] + [(s + n)(P0(m2) - P0(m1)) - (P1(m2) - P1(m1))]
            // (T1, s - 1 and s + n carry the factor r = 2^(k1-k0), see the header), plus the same sum of the zero row; a flat
            // window is a difference of P0 (+ the zero row's sum) over the gene count
            int lnan = 0;
            const unsigned tot_base = (unsigned)(kSeScratchOff + offsetof(ScratchE, tot));
            const unsigned lane4 = ((unsigned)tl & (kSeRep - 1)) * 4u;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                const int j = tl + i * NT;
                wvA[i] = 0.0;
                unsigned hb16 = 0xffffu;
                if (j < W) {
                    const unsigned w0 = wt[i].x, w1 = wt[i].y;
                    const double wbase = __hiloint2double((int)wt[i].w, (int)wt[i].z);
                    const f64x2 pb = ICV_LDS_D2_AT(kSePlanesOff + ((w0 & 0x1fffu) << 4));
                    const f64x2 pe = ICV_LDS_D2_AT(kSePlanesOff + ((w1 & 0x1fffu) << 4));
                    // (skipping the two total reads in wavefronts without a crossing window was measured: no gain)
                    const f64x2 te = ICV_LDS_D2_AT(tot_base + (((w1 >> 27) & 0xfu) << 4));
                    const double sgd = (double)(int)((w1 >> 13) & 0x3fffu);
                    double v;
                    if (__builtin_expect((int)w0 >= 0, 1)) {
                        const f64x2 pm = ICV_LDS_D2_AT(kSePlanesOff + (((w0 >> 13) & 0x1fffu) << 4));
                        const f64x2 tm = ICV_LDS_D2_AT(tot_base + (((w0 >> 26) & 0xfu) << 4));
                        const double pmx = pm.x + tm.x, pmy = pm.y + tm.y;
                        const double pex = pe.x + te.x, pey = pe.y + te.y;
                        const double sgm1 = fma(sgd, rr, -rr);  // (s - 1) r
                        const double sgn = sgm1 + win_r;        // (s + n) r
                        const double a = fma(-sgm1, pmx - pb.x, pmy - pb.y);
                        const double d = fma(sgn, pex - pmx, pmy - pey);
                        v = finish_window(fma(a + d, q1inv, wbase), 1, pyr_den, pyr_rcp, 1.0);
                    } else {  // flat: the gene count travels in the offset field
                        v = fma((pe.x + te.x) - pb.x, q0inv, wbase) / sgd;
                    }
                    wvA[i] = v;
                    lnan |= (v != v);
                    const int hb = hist_bin(v, inv_bound);
                    hb16 = (unsigned)hb;
                    // 16-bit fine bins, two per word; coarse bin hb >> 6, replica lane & 3: absolute LDS addresses
                    ICV_LDS_ADD_U32((unsigned)kSeHistOff + (((unsigned)hb >> 1) << 2), 1u << ((hb & 1) * 16));
                    ICV_LDS_ADD_U32((unsigned)kSeCoarseOff + ((((unsigned)hb >> 2) & 0x3f0u) | lane4), 1u);
                }
                wbA[i >> 1] = (i & 1) ? ((wbA[i >> 1] & 0xffffu) | (hb16 << 16)) : hb16 | 0xffff0000u;
            }
            if (lnan) sc->nanflag = 1;  // benign race: every writer stores 1
        }
        ICV_SEP(6)
        __syncthreads();  // B3: histograms of cell it complete, prefix sums consumed; median of cell it-1 published
        ICV_SEP(7)
    };

    // cells 0 .. n_mine-1, one iteration ahead to fill the load pipeline, two behind for the medians and x_res
    for (int64_t it = -1; it <= n_mine + 1; it += 2) {
        iteration(it, wvO, wbO, wvE, wbE);  // odd iteration: its cells live in the "odd" registers
        if (it + 1 <= n_mine + 1) iteration(it + 1, wvE, wbE, wvO, wbO);
    }
    if constexpr (CHUNK) {
        if (chunk_cur >= 0) {
            const double2 m2 = wave_moments(accS, accQ);
            if ((t & 63) == 0)
                reinterpret_cast<double2*>(P.chunk_part)[(chunk_cur * gridDim.x + blockIdx.x) * NWAVE + (t >> 6)] = m2;
        }
    }
#ifdef ICV_SE_PROFILE
    if (P.dbg && t == ICV_SE_PROFILE)
        for (int i = 0; i < 8; ++i) atomicAdd(P.dbg + i, sc->tacc[i]);
#endif
#undef ICV_SEP
}

}  // namespace icv

// k_smooth_x16: the dense float32 smoothing kernel as ONE 1024-thread workgroup (16 wavefronts) per CU.
//
// Round-2 measurements (profiles/r02_microbench_ops.txt): on gfx950 every VALU instruction -- float32 VOP3, SDWA,
// DPP, float64 alike -- costs ~4.7 cycles of a SIMD's issue (v_add_f32: 2.8), float64 MFMA shares the float64 VALU
// datapath, and the previous kernel issued ~5 800 VALU wave-instructions per cell: the smoothing kernel is bound by
// VALU INSTRUCTION COUNT, not by HBM and not by LDS.  This kernel is built around that:
//   * the reference row (20 VGPRs), the scatter table (10 VGPRs) and the window descriptors of a thread are loaded
//     ONCE per kernel: no per-cell table loads or address unpacking beyond one SDWA shift per gene;
//   * all 16 wavefronts run the same phase; row, {S0,S1} and the histograms are separate LDS regions (149 KB of
//     160 KB), nothing aliases, TWO barriers per cell;
//   * a thread sums two adjacent blocks per S pass (16-byte LDS reads) and the windows t, t + W/2 and two ADJACENT windows of one chromosome:
//     NBW + step / BT conflict-free {S0,S1} reads serve both windows;
//   * median: a 4096-bin histogram (32-bit LDS atomics) plus a 64-bin coarse histogram (4 replicas) -- ONE
//     wavefront resolves the two middle ranks with two 64-lane DPP prefix sums (coarse, then the 64 fine bins of
//     the located coarse bin), the windows of those bins are gathered, one wavefront ranks them exactly in
//     float64; all of it runs in the shadow of the next two cells (windows double buffered in registers);
//   * the row in LDS is being processed while the next one is in flight in registers (5 x 16 B per lane = 80 KB
//     per CU, re-requested vector by vector as it is consumed).
//
//   iteration `it` of a workgroup (cell = blockIdx.x + it * gridDim.x), p = it & 1:
//     phase A   S(it): LDS row -> {S0,S1} per block | wavefront 0: histogram scan(it-1) | wavefront 1: rank(it-2)
//     barrier 1
//     phase B   output(it-2) | gather(it-1), clear histogram(it-1) | W(it): windows, histogram atomics
//               | L(it+1): centre, clip, scatter the prefetched row; re-request the row of it+2
//     barrier 2
//
// Arithmetic and evaluation order of windows and median are those of k_smooth (bit-identical x_res and medians);
// the per-cell moments are reduced over 16 wavefront partials (last-bit differences in the sums).
// Geometry: float32 dense, one reference row, block form with compile-time block size, G <= 20 480 columns,
// blocks <= 2048, windows <= 2048.  Everything else runs k_smooth_ws / k_smooth.
#pragma once
#include "icv_kernel_ws.hpp"
#include "icv_plan.hpp"

namespace icv {

constexpr int XT = 1024;
constexpr int XWAVE = XT / 64;
constexpr int XU = 5;        // 16-byte row vectors per thread
// histogram: FINE bins (32-bit counters; 4096 for the window-100 geometry = the k_smooth_ws binning, 1024 where LDS
// is short) + FINE / 64 coarse bins (coarse = fine >> 6), each coarse bin in XREP replicas (lane & 15) adjacent in
// LDS: lanes of one atomic instruction that share a coarse bin hit different banks.  Both cell parities.
constexpr int XREP = 16;
constexpr int x16_hist_bytes(int fine) { return 2 * (fine * 4 + XREP * (fine / 64) * 4); }

struct ScratchX {
    int sel[2][8];  // located bins of the two middle ranks: b1, b2, below, c1, c2, nan
    int ncand[2];
    int nanflag[2];
    double med[2][2];  // the two middle order statistics
    double cand[2][64];
};
static_assert(sizeof(ScratchX) <= 1280, "ScratchX must fit the scratch region");

// both moments of a cell reduced over the wavefront at once: the first level moves the sum partials to lanes
// 0..31 and the sum-of-squares partials to lanes 32..63 (v_permlane32_swap), four DPP levels finish both.
// Fixed order -> deterministic.  Returns {sum, sq} (uniform).
__device__ __forceinline__ double2 wave_moments(double sum, double sq) {
    int a_lo = __double2loint(sum), a_hi = __double2hiint(sum);
    int b_lo = __double2loint(sq), b_hi = __double2hiint(sq);
    const auto l = __builtin_amdgcn_permlane32_swap(a_lo, b_lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(a_hi, b_hi, false, false);
    double v = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return make_double2(readlane_d(v, 0) + readlane_d(v, 16), readlane_d(v, 32) + readlane_d(v, 48));
}

// Monotone non-decreasing map window value -> fine bin (the k_smooth_ws binning scaled to FINE bins: the central
// quarter of [-bound, bound] gets 3/4 of the bins, each tail 1/8), scale folded into one FMA, floor + convert in one
// instruction (v_cvt_flr_i32_f32): 5 VALU for the central segment.
template <int FINE>
__device__ __forceinline__ int hist_bin_x(double v, float inv_bound, float c_scale, float c_thr) {
    constexpr int T = FINE / 8;  // bins per tail
    const float f = (float)v;
    if (__builtin_expect(fabsf(f) < c_thr, 1)) {  // |u| < 1/8: 3 FINE / 4 bins
        int b;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(b) : "v"(fmaf(f, c_scale, (float)(FINE / 2))));
        return b < T ? T : (b > FINE - T - 1 ? FINE - T - 1 : b);  // segments stay disjoint under rounding
    }
    float ct = c_thr;
    asm volatile("" : "+v"(ct));  // formed here, not hoisted: the (rare) tail path keeps no register alive
    const float u = f * (0.125f / ct);  // = f / bound
    (void)inv_bound;
    float g;
    int lo_b, hi_b;
    if (u < 0.0f) { g = (u + 1.0f) * ((float)T / 0.875f); lo_b = 0; hi_b = T - 1; }
    else { g = fmaf(u - 0.125f, (float)T / 0.875f, (float)(FINE - T)); lo_b = FINE - T; hi_b = FINE - 1; }
    const int b = (int)floorf(g);  // NaN -> 0 after the clamps; the cell is flagged separately
    return b < lo_b ? lo_b : (b > hi_b ? hi_b : b);
}

#ifndef ICV_X_WCH
#define ICV_X_WCH 2  // {S0,S1} pairs of a window read per batch (register budget: 5 spills next to the resident tables)
#endif
#ifndef ICV_X_ROW_AUX
#define ICV_X_ROW_AUX 2  // cache policy of the row loads: nt (the row is read once, by one CU): -3 % kernel time in alternating runs
#endif

// CHUNK: the moments of x_res are accumulated per thread over the consecutive cells of a noise-threshold chunk and
// reduced once per chunk (P.chunk_part), instead of one 16-wavefront reduction per cell (P.cell_part)
// REFRES: the reference row lives in registers for the whole kernel (20 VGPRs); false: re-read from L2 at the start
// of every L phase (long windows need the registers for larger batches of {S0,S1} reads)
// WIN: the float64 windows of every cell (before centring) also go to P.win_out[cell * P.win_ld + j] -- what
// calculate_gene_values averages (reference tl/_infercnv.py:274-288); cells handed back are rewritten by k_smooth
template <int BT, int NBW, int SB /* blocks between adjacent windows = step / BT */, bool CHUNK, int FINE, bool REFRES,
          bool WIN = false>
__global__ void __launch_bounds__(XT) k_smooth_x16(const KParams P) {
    constexpr int XFINE = FINE, XCOARSE = FINE / 64;
    static_assert(FINE % 1024 == 0 && XCOARSE * XREP <= XT && FINE / 4 <= XT, "histogram cleared by one store per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* row = reinterpret_cast<float*>(smem);
    double* S01 = reinterpret_cast<double*>(smem + P.win_off);
    unsigned* hist = reinterpret_cast<unsigned*>(smem + P.hist_off);  // [2][XFINE] fine, then [2][XREP][XCOARSE]
    unsigned* coarse = hist + 2 * XFINE;
    ScratchX* sc = reinterpret_cast<ScratchX*>(smem + P.scratch_off);
    float* stage = reinterpret_cast<float*>(smem + P.scratch_off + kFastScratchBytes);  // x_res of one cell (16-byte stores)
    static_assert(BT >= 2 && NBW > 0 && NBW % 2 == 0, "compile-time block size, even number of blocks per window");


    const int t = threadIdx.x;
    const int W = P.W, NB = P.NB;
    const int k1 = (W - 1) / 2, k2 = W / 2;
    // wavefront-uniform float constants, forced into SGPRs (they come out of float64 -> float32 conversions, which
    // would otherwise park them in VGPRs for the whole kernel)
    auto uniform = [](float x) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
    const float inv_bound = uniform((float)(1.0 / P.med_bound));
    const float c_scale = uniform(inv_bound * ((float)(3 * FINE / 4) / 0.25f)), c_thr = uniform(0.125f * (float)P.med_bound);
    const float cap = uniform((float)P.cap);
    const unsigned row_bytes = (unsigned)P.n_cols * 4u;
    const unsigned voff = (unsigned)t * 16u;
    const double pyr_den = P.pyr_den, pyr_rcp = P.pyr_rcp;
    const float* xbase = static_cast<const float*>(P.values);
    const int64_t n_mine = (P.n_rows - blockIdx.x + gridDim.x - 1) / gridDim.x;

    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(smem)) != 0)
        __builtin_trap();  // the L phase addresses the row by absolute LDS offsets

    // ---- per-thread constants, loaded once -------------------------------------------------------
    u32x4 refv[REFRES ? XU : 1];
    const __amdgpu_buffer_rsrc_t ref_rs = make_rsrc(P.ref_lo, row_bytes);
    unsigned laddr[XU][4];  // LDS byte addresses of the thread's 20 genes, resident for the whole kernel
    {
        const __amdgpu_buffer_rsrc_t d16_rs = make_rsrc(P.dst16, (unsigned)(XU * XT * 8));
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if constexpr (REFRES) refv[u] = __builtin_amdgcn_raw_buffer_load_b128(ref_rs, voff, u * XT * 16, 0);
            const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(d16_rs, (unsigned)t * 8u, u * XT * 8, 0);
            laddr[u][0] = (d.x & 0xffffu) * 4u;
            laddr[u][1] = (d.x >> 16) * 4u;
            laddr[u][2] = (d.y & 0xffffu) * 4u;
            laddr[u][3] = (d.y >> 16) * 4u;
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_X_EXP_LINSCAT)
            // upper-bound experiment (WRONG RESULTS; tools/build_variant.sh only): conflict-free scatter addresses
            laddr[u][0] = ((unsigned)(u * 4 + 0) * 1000u + (unsigned)t) * 4u;
            laddr[u][1] = ((unsigned)(u * 4 + 1) * 1000u + (unsigned)t) * 4u;
            laddr[u][2] = ((unsigned)(u * 4 + 2) * 1000u + (unsigned)t) * 4u;
            laddr[u][3] = ((unsigned)(u * 4 + 3) * 1000u + (unsigned)t) * 4u;
            if (t >= 1000) laddr[u][0] = laddr[u][1] = laddr[u][2] = laddr[u][3] = 80000u;
#endif
        }
    }
    // windows: thread t owns the adjacent windows j0, j0 + 1 of ONE chromosome (plan table x16_wdesc: first block,
    // first window, validity, "full pyramid window"), so the two windows share 9 of their 10 {S0,S1} pairs: 11 LDS
    // reads for both instead of 20.  {S0,S1} of block b lives at slot (b >> 1) + half (b & 1): the reads of a
    // wavefront (block stride 2 between lanes) and the writes of the S phase are 16-byte strided, conflict-free.
    const unsigned wdx = P.x16_wdesc[t];
    const int wb0 = (int)(wdx & 0xfffu), wj0 = (int)((wdx >> 12) & 0xfffu);
    const bool valid0 = (wdx >> 24) & 1u, valid1 = (wdx >> 25) & 1u;
    constexpr int INTER = 2 * SB;  // {S0,S1} of block b lives in array b % INTER at slot b / INTER
    const int half = P.x16_half;   // slots per array
    unsigned spK[INTER];           // LDS byte address of blocks b0, b0 + 1, ..., b0 + INTER - 1
#pragma unroll
    for (int k = 0; k < INTER; ++k)
        spK[k] = (unsigned)P.win_off + (unsigned)((wb0 + k) / INTER + half * ((wb0 + k) % INTER)) * 16u;
    const bool wave_w = __builtin_amdgcn_ballot_w64(valid0) != 0;  // the wavefront has windows at all
    const bool wfull = __builtin_amdgcn_ballot_w64((valid0 && !((wdx >> 26) & 1u)) || (valid1 && !((wdx >> 27) & 1u))) == 0;
    // pad slots and the trash slot are written once: nothing aliases the row
    for (int i = t; i < P.n_pad; i += XT) row[P.pad_idx[i]] = 0.0f;
    for (int i = t; i < x16_hist_bytes(FINE) / 4; i += XT) hist[i] = 0u;
    if (t < 2) {
        sc->ncand[t] = 0;
        sc->nanflag[t] = 0;
        sc->med[t][0] = 0.0;
        sc->med[t][1] = 0.0;
    }
    if (t < 16) sc->sel[t >> 3][t & 7] = 0;
    // x_res leaves the CU as 16-byte stores (store ISSUE, not bytes, is what a narrow store costs: -10 % kernel time
    // against two 4-byte stores per thread): needs 16-byte aligned rows
    const bool st16 = ((P.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.out) & 15) == 0);
    for (int i = t; i < ((W + 3) & ~3); i += XT) stage[i] = 0.0f;
    for (int i = t; i < 2 * INTER * half; i += XT) S01[i] = 0.0;  // slots past the last block are read (and discarded)

    u32x4 xq[XU];
    {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + (int64_t)blockIdx.x * P.ld, row_bytes);
#pragma unroll
        for (int u = 0; u < XU; ++u) xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * XT * 16, ICV_X_ROW_AUX);
    }

    // centre, clip and scatter the row in xq, re-requesting every vector for cell `c_next` as it is consumed
    auto l_phase = [&](int64_t c_next) __attribute__((always_inline)) {
        const bool more = c_next < P.n_rows;
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_X_EXP_NOLOAD)
        // upper-bound experiment (WRONG RESULTS; tools/build_variant.sh only): empty range, no HBM row traffic
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase, (more && c_next < 0) ? row_bytes : 0u);
#elif defined(ICV_DEV_EXPERIMENTS) && defined(ICV_X_EXP_L2ROW)
        // upper-bound experiment (WRONG RESULTS): every cell re-reads the workgroup's first row (L2 hits)
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + (int64_t)blockIdx.x * P.ld, more ? row_bytes : 0u);
#else
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + (more ? c_next : 0) * P.ld, more ? row_bytes : 0u);
#endif
        u32x4 rloc[REFRES ? 1 : XU];
        if constexpr (!REFRES) {
#pragma unroll
            for (int u = 0; u < XU; ++u) rloc[u] = __builtin_amdgcn_raw_buffer_load_b128(ref_rs, voff, u * XT * 16, 0);
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const u32x4 rv = REFRES ? refv[REFRES ? u : 0] : rloc[REFRES ? 0 : u];
            // two subtractions per instruction (v_pk_add_f32 with negated second operand)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 ya = f32x2{__uint_as_float(xq[u].x), __uint_as_float(xq[u].y)} -
                             f32x2{__uint_as_float(rv.x), __uint_as_float(rv.y)};
            const f32x2 yb = f32x2{__uint_as_float(xq[u].z), __uint_as_float(xq[u].w)} -
                             f32x2{__uint_as_float(rv.z), __uint_as_float(rv.w)};
            const float y0 = ya.x, y1 = ya.y, y2 = yb.x, y3 = yb.y;
            // v_med3 drops NaNs, np.clip keeps them: unordered pairs take the (never taken on real data) fix-up
            const bool un = __builtin_isunordered(y0, y1) | __builtin_isunordered(y2, y3);
            ICV_LDS_F32_AT(laddr[u][0]) = __builtin_amdgcn_fmed3f(y0, -cap, cap);
            ICV_LDS_F32_AT(laddr[u][1]) = __builtin_amdgcn_fmed3f(y1, -cap, cap);
            ICV_LDS_F32_AT(laddr[u][2]) = __builtin_amdgcn_fmed3f(y2, -cap, cap);
            ICV_LDS_F32_AT(laddr[u][3]) = __builtin_amdgcn_fmed3f(y3, -cap, cap);
            if (__builtin_expect(un, 0)) {
                if (y0 != y0) ICV_LDS_F32_AT(laddr[u][0]) = y0;
                if (y1 != y1) ICV_LDS_F32_AT(laddr[u][1]) = y1;
                if (y2 != y2) ICV_LDS_F32_AT(laddr[u][2]) = y2;
                if (y3 != y3) ICV_LDS_F32_AT(laddr[u][3]) = y3;
            }
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_X_EXP_NOLOAD)
            asm volatile("" : "+v"(xq[u]));  // the workgroup's first row stays in the registers: no row traffic at all
#else
            xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * XT * 16, ICV_X_ROW_AUX);  // out of range: zeros, no traffic
#endif
        }
    };

    // the first row is scattered before the loop; its successor is requested right away
    l_phase((int64_t)blockIdx.x + gridDim.x);

    // windows 2t, 2t+1 of cells of even / odd iteration (double buffered), and their fine bins (-1: no window)
    double wvE0 = 0.0, wvE1 = 0.0, wvO0 = 0.0, wvO1 = 0.0;
    unsigned wbE = 0xffffffffu, wbO = 0xffffffffu;  // two 16-bit bins per register (0xffff: no window)
    __syncthreads();

#ifdef ICV_X_PROFILE
#ifndef ICV_X_PROFILE_T
#define ICV_X_PROFILE_T 320
#endif
    // phase timers of one wavefront, accumulated in LDS (no registers held across the loop)
    unsigned long long* tacc = reinterpret_cast<unsigned long long*>(smem + P.scratch_off + 1280);
    if (t == ICV_X_PROFILE_T)
        for (int i = 0; i < 9; ++i) tacc[i] = 0;
#define ICV_XPH(i)                                              \
    if (P.dbg && t == ICV_X_PROFILE_T) {                        \
        unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tacc[i] += now_ - tacc[8];                              \
        tacc[8] = now_;                                         \
    }
    if (P.dbg && t == ICV_X_PROFILE_T) tacc[8] = __builtin_amdgcn_s_memtime();
#else
#define ICV_XPH(i)
#endif

    const int s_q = ((NB + 1) / 2 + XT - 1) / XT;  // S passes per wavefront on average (1: window 100, 2: window 250)
    // CHUNK: running moments of this thread's windows over the cells of the current chunk
    double accS = 0.0, accQ = 0.0;
    int64_t chunk_cur = -1, chunk_end = INT64_MIN;
    (void)accS;
    (void)accQ;
    (void)chunk_cur;
    (void)chunk_end;
    // one iteration; wvA/wbA: registers of cells with the parity of `it`, wvB/wbB: the other parity
    auto iteration = [&](int64_t it, int p0, double& wvA0, double& wvA1, unsigned& wbA, double& wvB0, double& wvB1,
                         unsigned& wbB) __attribute__((always_inline)) {
        const int p1 = p0 ^ 1;
        const int64_t cell = (int64_t)blockIdx.x + it * gridDim.x;
        const bool have0 = it < n_mine;                 // cell it: S, W
        const bool have1 = it >= 1 && it - 1 < n_mine;  // cell it-1: scan, gather
        const bool have2 = it >= 2 && it - 2 < n_mine;  // cell it-2: rank, output
        int tl = t;
        asm volatile("" : "+v"(tl));  // keep thread-derived addresses and predicates out of LICM (register budget)

        // =============================== phase A ================================================
        const int ts = tl - XT / 2;  // wavefronts 8..15 (wavefronts 0, 1 carry the median chains of this phase)
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_X_EXP_NOSTORE)
        if (st16 && it >= 3 && ts >= 0 && 4 * ts < W && P.dbg) {  // upper-bound experiment (WRONG RESULTS): no x_res stores
#else
        if (st16 && it >= 3 && ts >= 0 && 4 * ts < W) {
#endif
            // ---- x_res of cell it-3, staged in LDS by phase B of the previous iteration: one 16-byte store ----
            float* orow = P.out + (cell - 3 * (int64_t)gridDim.x) * P.ldo;
            const float4 q = *reinterpret_cast<const float4*>(stage + 4 * ts);
            if (4 * ts + 4 <= W) {
                {
                    // non-temporal: x_res is written once and never read by this kernel; measured -2 % kernel time in
                    // alternating runs against a plain store (the stores cost clock)
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    const f32x4_t qq = {q.x, q.y, q.z, q.w};
                    __builtin_nontemporal_store(qq, reinterpret_cast<f32x4_t*>(orow + 4 * ts));
                }
            } else {
                orow[4 * ts] = q.x;
                if (4 * ts + 1 < W) orow[4 * ts + 1] = q.y;
                if (4 * ts + 2 < W) orow[4 * ts + 2] = q.z;
            }
        }
        ICV_XPH(6)
        // the two single-wavefront chains (LDS round trips, DPP scans, lane reads: ~1000 cycles of latency each) run
        // on wavefronts 0 and 1, which take no part in the S phase; their SIMDs' share of the block sums goes to
        // wavefronts 4 and 5 (two passes), so every SIMD issues four S passes
        if (have1 && tl < 64) {
            // ---- wavefront 0: the bins of the two middle ranks of cell it-1 -----------------------
            // (all LDS reads of a step are issued together: one round trip, not one per use)
            const unsigned* cz = coarse + p1 * (XREP * XCOARSE);
            const unsigned* fz = hist + p1 * XFINE;
            const uint4* c4 = reinterpret_cast<const uint4*>(cz + (tl < XCOARSE ? tl : 0) * XREP);
            const int nanf = sc->nanflag[p1];
            const uint4 ca = c4[0], cb = c4[1], cc = c4[2], cd = c4[3];
            if (nanf) {
                if (tl == 0) {
                    sc->nanflag[p1] = 0;
                    sc->sel[p1][5] = 1;
                }
            } else {
                const int csum = (int)((((ca.x + ca.y) + (ca.z + ca.w)) + ((cb.x + cb.y) + (cb.z + cb.w))) +
                                       (((cc.x + cc.y) + (cc.z + cc.w)) + ((cd.x + cd.y) + (cd.z + cd.w))));
                const int c = tl < XCOARSE ? csum : 0;
                const int cincl = wave_scan_dpp(c);
                int Cprev = -1, f = 0, fincl = 0;
                int res[2][3];
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const int k = which == 0 ? k1 : k2;
                    const int C = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(cincl > k));
                    const int belowC = __builtin_amdgcn_readlane(cincl - c, C);
                    if (C != Cprev) {  // wavefront-uniform
                        f = (int)fz[C * 64 + tl];
                        fincl = wave_scan_dpp(f);
                        Cprev = C;
                    }
                    const int j = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(fincl + belowC > k));
                    res[which][0] = C * 64 + j;
                    res[which][1] = __builtin_amdgcn_readlane(fincl - f, j) + belowC;
                    res[which][2] = __builtin_amdgcn_readlane(f, j);
                }
                if (tl == 0) {
                    *reinterpret_cast<int4*>(sc->sel[p1]) = make_int4(res[0][0], res[1][0], res[0][1], res[0][2]);
                    *reinterpret_cast<int2*>(sc->sel[p1] + 4) = make_int2(res[1][2], 0);
                }
            }
        }
        if (have2 && (tl >> 6) == 1) {
            // ---- wavefront 1: exact float64 ranks of the <= 64 candidates of cell it-2 -------------
            const int lane = tl & 63;
            const int4 s = *reinterpret_cast<const int4*>(sc->sel[p0]);       // b1, b2, below, c1
            const int2 s2 = *reinterpret_cast<const int2*>(sc->sel[p0] + 4);  // c2, nan
            const int ncr = sc->ncand[p0];
            const double mine_raw = sc->cand[p0][lane];  // unconditional: one LDS round trip for all four reads
            const int nin = s.w + (s.y != s.x ? s2.x : 0);
            double ma = 0.0, mb = 0.0;  // handed-back cell: placeholder, rewritten by k_smooth
            if (s2.y) {
                ma = mb = __builtin_nan("");
            } else if (nin <= 64) {
                const int n = ncr < 64 ? ncr : 64;
                const double mine = (lane < n) ? mine_raw : __builtin_inf();
                int r = 0;
                for (int q = 0; q < n; ++q) {  // n is wavefront-uniform (typically 2..4)
                    const double o = readlane_d(mine, q);
                    r += (int)(o < mine) | ((int)(o == mine) & (int)(q < lane));
                }
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(lane < n && r == k1 - s.z);
                const unsigned long long m2 = __builtin_amdgcn_ballot_w64(lane < n && r == k2 - s.z);
                ma = readlane_d(mine, m1 ? (int)__builtin_ctzll(m1) : 0);
                mb = readlane_d(mine, m2 ? (int)__builtin_ctzll(m2) : 0);
            }
            if (lane == 0) {
                *reinterpret_cast<double2*>(sc->med[p0]) = make_double2(ma, mb);
                sc->ncand[p0] = 0;
            }
        }
        ICV_XPH(7)
        if (have0) {
            // ---- S: block partial sums, straight into their own LDS region.  A pass of a wavefront covers 64
            // pairs of adjacent blocks.  With q = ceil(pairs / 1024) passes per wavefront on average, the chain
            // wavefronts 0, 1 take q - 1 passes, wavefronts 4, 5 (same SIMDs) q + 1, the others q: every SIMD
            // issues 4 q passes.  Pass k of wavefront w covers slot base(w) + k (slots dealt in wavefront order).
            const int wv_id = __builtin_amdgcn_readfirstlane(tl >> 6);
            const int q = s_q;
            const int n_pass = wv_id < 2 ? q - 1 : ((wv_id == 4 || wv_id == 5) ? q + 1 : q);
            // slots before wavefront w: w * q - min(w, 2) + (w > 4) + (w > 5)
            const int slot0 = wv_id * q - (wv_id < 2 ? wv_id : 2) + (wv_id > 4 ? 1 : 0) + (wv_id > 5 ? 1 : 0);
            for (int pass = 0; pass < n_pass; ++pass) {
                const int b = 2 * ((slot0 + pass) * 64 + (tl & 63));
                if (b < NB) {
                    float v[2 * BT];
                    if constexpr ((2 * BT * 4) % 16 == 0) {  // 16-byte aligned pairs of blocks
                        const float4* rp = reinterpret_cast<const float4*>(row + b * BT);
#pragma unroll
                        for (int r = 0; r < 2 * BT / 4; ++r) {
                            const float4 qv = rp[r];
                            v[4 * r] = qv.x;
                            v[4 * r + 1] = qv.y;
                            v[4 * r + 2] = qv.z;
                            v[4 * r + 3] = qv.w;
                        }
                    } else {  // 8-byte aligned (BT = 5: 40 bytes per pair)
                        const float2* rp = reinterpret_cast<const float2*>(row + b * BT);
#pragma unroll
                        for (int r = 0; r < BT; ++r) {
                            const float2 qv = rp[r];
                            v[2 * r] = qv.x;
                            v[2 * r + 1] = qv.y;
                        }
                    }
                    // canonical order of block_accumulate without its two no-ops (0 + v0, fma(0, v0, 0) and
                    // fma(1, v1, 0)): same values, only the sign of an all-zero sum can differ
                    double s0a = (double)v[0], s0b = (double)v[BT];
                    double s1a, s1b;
                    {
                        const double a1 = (double)v[1], b1 = (double)v[BT + 1];
                        s0a = s0a + a1;
                        s0b = s0b + b1;
                        s1a = a1;
                        s1b = b1;
                    }
#pragma unroll
                    for (int r = 2; r < BT; ++r) {
                        block_accumulate((double)v[r], r, s0a, s1a);
                        block_accumulate((double)v[BT + r], r, s0b, s1b);
                    }
                    double2* sp = reinterpret_cast<double2*>(S01);  // INTER arrays of `half` slots
                    sp[b / INTER + half * (b % INTER)] = make_double2(s0a, s1a);
                    if (b + 1 < NB) sp[(b + 1) / INTER + half * ((b + 1) % INTER)] = make_double2(s0b, s1b);
                }
            }
        }
        ICV_XPH(0)
        __syncthreads();  // barrier 1: {S0,S1} of cell it complete, row dead; bins / median published
        ICV_XPH(1)
        asm volatile("" : "+v"(tl));

        // =============================== phase B ================================================
        // ---- x_res of cell it-2 from its windows (still in wvA): staged in LDS (stored as 16-byte vectors by phase
        // A of the next iteration) and its moments
        float yf0 = 0.0f, yf1 = 0.0f;
        double2 mo = make_double2(0.0, 0.0);
        double med_out = 0.0;
        if (have2) {
            const int64_t pcell = cell - 2 * (int64_t)gridDim.x;
            const double2 mm = *reinterpret_cast<const double2*>(sc->med[p0]);
            med_out = (k1 == k2) ? mm.x : (mm.x + mm.y) / 2.0;
            if constexpr (CHUNK) {
                if (pcell >= chunk_end) {  // uniform: first cell of this workgroup in a new chunk
                    if (wave_w && chunk_cur >= 0) {
                        const double2 m2 = wave_moments(accS, accQ);
                        if ((tl & 63) == 0)
                            reinterpret_cast<double2*>(P.chunk_part)[(chunk_cur * gridDim.x + blockIdx.x) * XWAVE + (tl >> 6)] = m2;
                    }
                    chunk_cur = (pcell + P.row_phase) / P.chunksize;
                    chunk_end = (chunk_cur + 1) * P.chunksize - P.row_phase;
                    accS = 0.0;
                    accQ = 0.0;
                }
            }
            if (wave_w) {
                const double y0 = wvA0 - med_out, y1 = wvA1 - med_out;
                const bool v0 = valid0, v1 = valid1;
                yf0 = (float)y0;
                yf1 = (float)y1;
                if constexpr (CHUNK) {
                    // a cell handed back to k_smooth (more than 64 windows in the median bins) is accounted there
                    const int4 s = *reinterpret_cast<const int4*>(sc->sel[p0]);
                    const int2 s2 = *reinterpret_cast<const int2*>(sc->sel[p0] + 4);
                    const bool handed = !s2.y && s.w + (s.y != s.x ? s2.x : 0) > 64;
                    if (!handed) {
                        if (v0) {
                            accS = accS + y0;
                            accQ = fma(y0, y0, accQ);
                        }
                        if (v1) {
                            accS = accS + y1;
                            accQ = fma(y1, y1, accQ);
                        }
                    }
                } else {
                    double sum = y0, sq = y0 * y0;
                    if (v1) {
                        sum = sum + y1;
                        sq = fma(y1, y1, sq);
                    }
                    if (!v0) sum = sq = 0.0;
                    mo = wave_moments(sum, sq);
                }
                if (st16) {
                    int js = wj0;
                    asm volatile("" : "+v"(js));  // address formed here, not held in a register across the loop
                    if (v0) stage[js] = yf0;
                    if (v1) stage[js + 1] = yf1;
                }
                if constexpr (WIN) {
                    double* wrow = P.win_out + pcell * P.win_ld;
                    if (v0) __builtin_nontemporal_store(wvA0, wrow + wj0);  // (written once, read by another kernel)
                    if (v1) __builtin_nontemporal_store(wvA1, wrow + wj0 + 1);
                }
            }
        }
        if (have1) {
            // ---- windows of cell it-1 (wvB) in the bins of its two middle ranks -> cand[]; no window lies in a
            // bin strictly between the bins of two adjacent ranks, so "b1 <= bin <= b2" selects exactly those bins
            const int4 s = *reinterpret_cast<const int4*>(sc->sel[p1]);
            const int2 s2 = *reinterpret_cast<const int2*>(sc->sel[p1] + 4);
            const int nin = s.w + (s.y != s.x ? s2.x : 0);
            if (!s2.y) {
                if (nin <= 64) {
                    const unsigned span = (unsigned)(s.y - s.x);
                    const bool hA = (wbB & 0xffffu) - (unsigned)s.x <= span, hB = (wbB >> 16) - (unsigned)s.x <= span;
                    if (__builtin_amdgcn_ballot_w64(hA | hB)) {
                        if (hA) {
                            const int idx = atomicAdd(&sc->ncand[p1], 1);
                            if (idx < 64) sc->cand[p1][idx] = wvB0;
                        }
                        if (hB) {
                            const int idx = atomicAdd(&sc->ncand[p1], 1);
                            if (idx < 64) sc->cand[p1][idx] = wvB1;
                        }
                    }
                } else if (tl == 0) {
                    // too many windows share the median bins: the generic kernel recomputes the cell
                    const int slot = atomicAdd(P.row_count, 1);
                    P.row_list[slot] = cell - (int64_t)gridDim.x;
                }
            }
            // clear the histograms of cell it-1 (scanned before barrier 1; next used by W of cell it+1)
            unsigned zero = 0u;
            asm volatile("" : "+v"(zero));  // materialised here (one v_mov), not four registers held across the loop
            if (tl < XFINE / 4) reinterpret_cast<uint4*>(hist + p1 * XFINE)[tl] = make_uint4(zero, zero, zero, zero);
            if (tl < XREP * XCOARSE) coarse[p1 * (XREP * XCOARSE) + tl] = zero;
        }
        ICV_XPH(4)
        if (have0 && wave_w) {
            // ---- W: windows 2t, 2t+1 of cell it from {S0,S1}, histogram atomics --------------------
            double v0, v1;
            auto fast_windows = [&]() __attribute__((always_inline))
            {
                // full pyramid windows, no per-window branches (the few flat windows -- one per chromosome with at
                // most `window` genes -- are overwritten below): window j0 uses
                // the pairs of blocks b0 .. b0+9, window j0+1 those of b0+1 .. b0+10 (canonical order inside a
                // window, the two float64 chains interleaved)
                constexpr int HB = NBW / 2;
#ifdef ICV_X_WCH_LONG
                constexpr int WCH = NBW > 10 ? ICV_X_WCH_LONG : ICV_X_WCH;
#else
                constexpr int WCH = NBW > 10 ? 5 : ICV_X_WCH;  // long windows: fewer, larger batches of LDS reads
#endif
                auto blk = [&](int i) __attribute__((always_inline)) {  // {S0,S1} of block b0 + i
                    return reinterpret_cast<const double2*>(smem + spK[i % INTER])[i / INTER];
                };
                v0 = 0.0;
                v1 = 0.0;
                double2 ring[SB + 1];  // blocks b0 + m .. b0 + m + SB
#pragma unroll
                for (int k = 0; k <= SB; ++k) ring[k] = blk(k);
#pragma unroll
                for (int m = 0; m < NBW; ++m) {
                    const double2 a = ring[m % (SB + 1)], b = ring[(m + SB) % (SB + 1)];
                    if (m + 1 < NBW) ring[m % (SB + 1)] = blk(m + SB + 1);  // consumed slot <- the block after the ring
                    const double wgt = (double)(m < HB ? m * BT + 1 : NBW * BT - m * BT);
                    v0 = fma(wgt, a.x, v0);  // window j0, its block m     = block b0 + m
                    v1 = fma(wgt, b.x, v1);  // window j0 + 1, its block m = block b0 + SB + m
                    v0 = m < HB ? v0 + a.y : v0 - a.y;
                    v1 = m < HB ? v1 + b.y : v1 - b.y;
                    if (m % WCH == WCH - 1) __builtin_amdgcn_sched_barrier(0);  // bound the reads in flight
                }
                v0 = finish_window(v0, NBW * BT, pyr_den, pyr_rcp, 1.0);
                v1 = finish_window(v1, NBW * BT, pyr_den, pyr_rcp, 1.0);
            };
            // (two call sites on purpose: with the fast path behind the wavefront-uniform test the register
            // allocator keeps the row prefetch in registers; as one unconditional block it spilled it)
            if (wfull) fast_windows();
            else fast_windows();
            if (!wfull) {
                // flat windows in this wavefront (one per chromosome with at most `window` genes; always the first
                // window of its thread's pair): plain mean of the chromosome's genes = sum of the S0 of its blocks in
                // block order (canonical), over the gene count.  Handled one at a time with wavefront-uniform
                // values: block count and gene count come through the SCALAR cache (a vector load here would drain
                // the row prefetch), the LDS reads are broadcasts, the owning lane keeps the result.
                unsigned long long fm = __builtin_amdgcn_ballot_w64(valid0 && !((wdx >> 26) & 1u));
                const int wbase = __builtin_amdgcn_readfirstlane(tl & ~63);
#pragma unroll 1
                while (fm) {
                    const int L = (int)__builtin_ctzll(fm);
                    fm &= fm - 1;
                    const unsigned fi = P.x16_wdesc[XT + wbase + L];  // uniform address: s_load
                    const int nb = (int)(fi & 0xffffu);
                    const int bs = __builtin_amdgcn_readlane(wb0, L);
                    const double2* sb = reinterpret_cast<const double2*>(S01);
                    // one LDS read per block, all in flight at once (lane m: block m), then the canonical
                    // left-to-right sum through lane reads
                    double acc = 0.0;
#pragma unroll 1
                    for (int base = 0; base < nb; base += 64) {
                        const int mm = base + (tl & 63);
                        const int bb = bs + (mm < nb ? mm : 0);
                        const double s0l = sb[bb / INTER + half * (bb % INTER)].x;
                        const int cnt = nb - base < 64 ? nb - base : 64;
#pragma unroll 1
                        for (int q = 0; q < cnt; ++q) acc = acc + readlane_d(s0l, q);
                    }
                    const double vf = acc / (double)(int)(fi >> 16);
                    if ((tl & 63) == L) v0 = vf;
                }
            }
            wvA0 = v0;
            wvA1 = v1;
            const int h0 = hist_bin_x<FINE>(v0, inv_bound, c_scale, c_thr), h1 = hist_bin_x<FINE>(v1, inv_bound, c_scale, c_thr);
            const bool w0 = valid0, w1 = valid1;
            wbA = (unsigned)(w0 ? h0 : 0xffff) | ((unsigned)(w1 ? h1 : 0xffff) << 16);
            unsigned* fz = hist + p0 * XFINE;
            unsigned* cz = coarse + p0 * (XREP * XCOARSE) + (tl & (XREP - 1));
            if (w0) {
                atomicAdd(fz + h0, 1u);
                atomicAdd(cz + (h0 >> 6) * XREP, 1u);
            }
            if (w1) {
                atomicAdd(fz + h1, 1u);
                atomicAdd(cz + (h1 >> 6) * XREP, 1u);
            }
            const double vs = w1 ? v0 + v1 : v0;  // NaN in either window (|window| is bounded: no inf - inf)
            if (w0 && vs != vs) sc->nanflag[p0] = 1;  // benign race: every writer stores 1
        }
        ICV_XPH(5)
        const int64_t nxt = cell + gridDim.x;
        if (nxt < P.n_rows) l_phase(nxt + gridDim.x);
        if (have2) {
            const int64_t pcell = cell - 2 * (int64_t)gridDim.x;
            if (!st16) {  // unaligned result rows: 4-byte stores, after the row loads of the L phase
                float* orow = P.out + pcell * P.ldo;
                if (valid0) orow[wj0] = yf0;
                if (valid1) orow[wj0 + 1] = yf1;
            }
            if constexpr (!CHUNK)
                if ((tl & 63) == 0) reinterpret_cast<double2*>(P.cell_part)[pcell * XWAVE + (tl >> 6)] = mo;
            if (tl == 0) P.cell_median[pcell] = med_out;
        }
        ICV_XPH(2)
        __syncthreads();  // barrier 2: row of cell it+1 scattered, histograms of cell it complete, {S0,S1} dead
        ICV_XPH(3)
    };

    const int64_t n_it = n_mine + 3;  // two iterations of median pipeline + one for the staged x_res store
    for (int64_t it = 0; it < n_it; it += 2) {
        iteration(it, 0, wvE0, wvE1, wbE, wvO0, wvO1, wbO);
        if (it + 1 < n_it) iteration(it + 1, 1, wvO0, wvO1, wbO, wvE0, wvE1, wbE);
    }
    if constexpr (CHUNK) {
        if (wave_w && chunk_cur >= 0) {
            const double2 m2 = wave_moments(accS, accQ);
            if ((t & 63) == 0)
                reinterpret_cast<double2*>(P.chunk_part)[(chunk_cur * gridDim.x + blockIdx.x) * XWAVE + (t >> 6)] = m2;
        }
    }
#ifdef ICV_X_PROFILE
    if (P.dbg && t == ICV_X_PROFILE_T)
        for (int i = 0; i < 8; ++i) atomicAdd(P.dbg + i, tacc[i]);
#endif
#undef ICV_XPH
}

// cell_stats[c] = sum over the n_part per-wavefront partial moment pairs of cell c, fixed order (deterministic)
__global__ void __launch_bounds__(256) k_stats_finish_n(const double* part, int64_t n_rows, int n_part,
                                                        double* cell_stats) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_rows) return;
    const double2* p = reinterpret_cast<const double2*>(part) + c * n_part;
    double s = 0.0, q = 0.0;
    for (int w = 0; w < n_part; ++w) {
        const double2 v = p[w];
        s += v.x;
        q += v.y;
    }
    cell_stats[2 * c] = s;
    cell_stats[2 * c + 1] = q;
}

}  // namespace icv

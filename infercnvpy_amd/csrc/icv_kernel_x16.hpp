// k_smooth_x16: the dense float32 smoothing kernel as ONE 1024-thread workgroup (16 wavefronts) per CU.
//
// Why (round-1 measurements): with two 512-thread workgroups per CU the cell time was the serial
// chain of one workgroup (S, W, median, output: latency-bound at 8 wavefronts) and the L phase was five L2 round
// trips for the reference row and the scatter table.  Here
//   * all 16 wavefronts run the same phase, so every phase is bound by a pipe (float64 VALU, LDS), not by latency;
//   * the reference row (20 VGPRs), the scatter table (10 VGPRs) and the window descriptors (2 VGPRs) of a thread
//     are loaded ONCE per kernel: no table traffic per cell, the only VMEM loads of the loop are the row prefetch;
//   * the row being processed lives in LDS, the next one is in flight in registers (5 x 16 B per lane = 80 KB per
//     CU, re-requested vector by vector as it is consumed);
//   * row, {S0,S1}, histogram are separate LDS regions (122 KB of 160 KB): nothing aliases, TWO barriers per cell;
//   * the median of a cell is resolved in the slack of the next three cells (windows triple buffered in
//     registers): histogram scan (A of cell+1), bin location (B of cell+1), candidate gather (A of cell+2), exact
//     float64 ranking by one wavefront (B of cell+2), x_res output (A of cell+3).
//
//   iteration `it` of a workgroup (cell k = blockIdx.x + it * gridDim.x):
//     phase A   output(it-3) | gather(it-2) | histogram scan(it-1) | S(it): LDS row -> {S0,S1} per block
//     barrier 1
//     phase B   rank(it-2) | locate(it-1) | W(it): windows from {S0,S1}, histogram atomics
//               | L(it+1): centre, clip, scatter the prefetched row; re-request the row of it+2
//     barrier 2
//
// Arithmetic and evaluation order of windows and median are those of k_smooth (bit-identical x_res and medians);
// the per-cell moments are reduced over 16 wavefront partials instead of 8 (last-bit differences in the sums).
// Geometry: float32 dense, one reference row, block form with compile-time block size, G <= 20 480 columns,
// blocks <= MAXB * 1024, windows <= MAXW * 1024.  Everything else runs k_smooth_ws / k_smooth.
#pragma once
#include "icv_kernel_ws.hpp"

namespace icv {

constexpr int XT = 1024;
constexpr int XWAVE = XT / 64;
constexpr int XU = 5;  // 16-byte row vectors per thread

struct ScratchX {
    int wtot[2][XWAVE];  // histogram scan: windows in the 256 bins of each wavefront, by cell parity
    int sel[2][8];       // located bins of the two middle ranks: b1, b2, below, c1, c2, nan
    int ncand[2];
    int nanflag[2];
    double med[2][2];    // the two middle order statistics
    double cand[2][64];
};
static_assert(sizeof(ScratchX) <= 1536, "ScratchX must fit the scratch region");

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

#ifndef ICV_X_WFIRST
#define ICV_X_WFIRST 1  // phase B order: 1 = every wavefront W then L; 2 = odd wavefronts L then W (pipes mixed)
#endif
#ifndef ICV_X_WCH
#define ICV_X_WCH 5  // {S0,S1} pairs of a window read per batch
#endif

template <int MAXB, int MAXW, int BT, int NBW>
__global__ void __launch_bounds__(XT) k_smooth_x16(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* row = reinterpret_cast<float*>(smem);
    double* S01 = reinterpret_cast<double*>(smem + P.win_off);
    int* hist = reinterpret_cast<int*>(smem + P.hist_off);
    ScratchX* sc = reinterpret_cast<ScratchX*>(smem + P.scratch_off);
    static_assert(NBIN == 4 * XT, "the histogram scan gives every thread 4 bins");
    static_assert(BT > 0 && NBW > 0 && NBW % 2 == 0, "compile-time block size and blocks per window");

    const int t = threadIdx.x;
    const int W = P.W, NB = P.NB;
    const int k1 = (W - 1) / 2, k2 = W / 2;
    const float inv_bound = (float)(1.0 / P.med_bound);
    const float cap = (float)P.cap;
    const unsigned row_bytes = (unsigned)P.n_cols * 4u;
    const unsigned voff = (unsigned)t * 16u;
    const double pyr_den = P.pyr_den, pyr_rcp = P.pyr_rcp;
    const float* xbase = static_cast<const float*>(P.values);
    const int64_t n_mine = (P.n_rows - blockIdx.x + gridDim.x - 1) / gridDim.x;

    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(smem)) != 0)
        __builtin_trap();  // the L phase addresses the row by absolute LDS offsets

    // ---- per-thread constants, loaded once -------------------------------------------------------
    u32x4 refv[XU];
    u32x2 dtab[XU];
    {
        const __amdgpu_buffer_rsrc_t lo_rs = make_rsrc(P.ref_lo, row_bytes);
        const __amdgpu_buffer_rsrc_t d16_rs = make_rsrc(P.dst16, (unsigned)(XU * XT * 8));
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            refv[u] = __builtin_amdgcn_raw_buffer_load_b128(lo_rs, voff, u * XT * 16, 0);
            dtab[u] = __builtin_amdgcn_raw_buffer_load_b64(d16_rs, (unsigned)t * 8u, u * XT * 8, 0);
        }
    }
    int wdesc[MAXW];   // (start block) | (len << 16); a thread's missing window repeats its first one
    bool wfull = true; // every window of this wavefront is a full pyramid window
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
        const int j = t + i * XT;
        wdesc[i] = P.w_pack[j < W ? j : (t < W ? t : 0)];
        wfull &= (wdesc[i] >> 16) == NBW * BT;
    }
    wfull = __builtin_amdgcn_ballot_w64(!wfull) == 0;
    // pad slots and the trash slot are written once: nothing aliases the row
    for (int i = t; i < P.n_pad; i += XT) row[P.pad_idx[i]] = 0.0f;
    for (int i = t; i < NBIN / 2; i += XT) hist[i] = 0;
    if (t < 2) {
        sc->ncand[t] = 0;
        sc->nanflag[t] = 0;
        sc->med[t][0] = 0.0;
        sc->med[t][1] = 0.0;
    }
    if (t < 16) sc->sel[t >> 3][t & 7] = 0;

    u32x4 xq[XU];
    {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + (int64_t)blockIdx.x * P.ld, row_bytes);
#pragma unroll
        for (int u = 0; u < XU; ++u) xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * XT * 16, 0);
    }
    unsigned two = 2u;
    asm volatile("" : "+v"(two));  // a VGPR operand for the SDWA shifts

    // centre, clip and scatter the row in xq (cell `c_row`), then re-request every vector for cell `c_next`
    auto l_phase = [&](int64_t c_next) __attribute__((always_inline)) {
        const bool more = c_next < P.n_rows;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + (more ? c_next : 0) * P.ld, more ? row_bytes : 0u);
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const float y0 = __uint_as_float(xq[u].x) - __uint_as_float(refv[u].x);
            const float y1 = __uint_as_float(xq[u].y) - __uint_as_float(refv[u].y);
            const float y2 = __uint_as_float(xq[u].z) - __uint_as_float(refv[u].z);
            const float y3 = __uint_as_float(xq[u].w) - __uint_as_float(refv[u].w);
            // v_med3 drops NaNs, np.clip keeps them: unordered pairs take the (never taken on real data) fix-up
            const bool un = __builtin_isunordered(y0, y1) | __builtin_isunordered(y2, y3);
            ICV_LDS_F32_AT(lds_off_lo16(dtab[u].x, two)) = __builtin_amdgcn_fmed3f(y0, -cap, cap);
            ICV_LDS_F32_AT(lds_off_hi16(dtab[u].x, two)) = __builtin_amdgcn_fmed3f(y1, -cap, cap);
            ICV_LDS_F32_AT(lds_off_lo16(dtab[u].y, two)) = __builtin_amdgcn_fmed3f(y2, -cap, cap);
            ICV_LDS_F32_AT(lds_off_hi16(dtab[u].y, two)) = __builtin_amdgcn_fmed3f(y3, -cap, cap);
            if (__builtin_expect(un, 0)) {
                unsigned dx = dtab[u].x, dy = dtab[u].y;
                asm volatile("" : "+v"(dx), "+v"(dy));  // keep the unpacked addresses out of LICM (20 VGPRs)
                if (y0 != y0) row[dx & 0xffffu] = y0;
                if (y1 != y1) row[dx >> 16] = y1;
                if (y2 != y2) row[dy & 0xffffu] = y2;
                if (y3 != y3) row[dy >> 16] = y3;
            }
            xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * XT * 16, 0);  // out of range: zeros, no traffic
        }
    };

    // the first row is scattered before the loop; its successor is requested right away
    l_phase((int64_t)blockIdx.x + gridDim.x);

    double wv0[MAXW], wv1[MAXW], wv2[MAXW];  // windows of cells it, it-1, it-2 (after the rotation in phase B)
    unsigned wb0 = 0, wb1 = 0;               // their histogram bins, 16 bits each (MAXW == 2)
    static_assert(MAXW == 2, "two windows per thread: bins packed in one register");
#pragma unroll
    for (int i = 0; i < MAXW; ++i) wv0[i] = wv1[i] = wv2[i] = 0.0;
    __syncthreads();

#ifdef ICV_X_PROFILE
    unsigned long long tlast = 0, tacc[4] = {0, 0, 0, 0};
#define ICV_XPH(i)                                              \
    if (P.dbg && t == 64) {                                     \
        unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tacc[i] += now_ - tlast;                                \
        tlast = now_;                                           \
    }
    if (P.dbg && t == 64) tlast = __builtin_amdgcn_s_memtime();
#else
#define ICV_XPH(i)
#endif

    for (int64_t it = 0; it < n_mine + 3; ++it) {
        const int64_t cell = (int64_t)blockIdx.x + it * gridDim.x;
        const bool have0 = it < n_mine;                      // cell it: S and W
        const bool have1 = it >= 1 && it - 1 < n_mine;       // cell it-1: scan, locate
        const bool have2 = it >= 2 && it - 2 < n_mine;       // cell it-2: gather, rank
        const bool have3 = it >= 3;                          // cell it-3: output
        const int p0 = (int)(it & 1), p1 = p0 ^ 1;           // parity of cells it / it-2, and it-1 / it-3
        int tl = t;
        asm volatile("" : "+v"(tl));  // keep thread-derived addresses and predicates out of LICM (register budget)

        // =============================== phase A ================================================
        if (have3) {
            // ---- x_res of cell it-3 from its windows (wv2), moments, median --------------------
            const int64_t pcell = cell - 3 * (int64_t)gridDim.x;
            const double2 mm = *reinterpret_cast<const double2*>(sc->med[p1]);
            const double med = (k1 == k2) ? mm.x : (mm.x + mm.y) / 2.0;
            double sum = 0.0, sq = 0.0;
            float* orow = P.out + pcell * P.ldo;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                const int j = tl + i * XT;
                if (j < W) {
                    const double y = wv2[i] - med;
                    orow[j] = (float)y;
                    sum = sum + y;
                    sq = fma(y, y, sq);
                }
            }
            const double2 mo = wave_moments(sum, sq);
            if ((tl & 63) == 0) reinterpret_cast<double2*>(P.cell_part)[pcell * XWAVE + (tl >> 6)] = mo;
            if (tl == 0) P.cell_median[pcell] = med;
        }
        if (have2) {
            // ---- windows of cell it-2 (wv1) in the bins of its two middle ranks -> cand[] ------
            const int4 s = *reinterpret_cast<const int4*>(sc->sel[p0]);  // b1, b2, below, c1
            const int2 s2 = *reinterpret_cast<const int2*>(sc->sel[p0] + 4);  // c2, nan
            const int nin = s.w + (s.y != s.x ? s2.x : 0);
            if (!s2.y) {
                if (nin <= 64) {
                    const unsigned bA = wb1 & 0xffffu, bB = wb1 >> 16;
                    const bool hA = (tl < W) & ((int)bA == s.x | (int)bA == s.y);
                    const bool hB = (tl + XT < W) & ((int)bB == s.x | (int)bB == s.y);
                    if (__builtin_amdgcn_ballot_w64(hA | hB)) {
                        if (hA) {
                            const int idx = atomicAdd(&sc->ncand[p0], 1);
                            if (idx < 64) sc->cand[p0][idx] = wv1[0];
                        }
                        if (hB) {
                            const int idx = atomicAdd(&sc->ncand[p0], 1);
                            if (idx < 64) sc->cand[p0][idx] = wv1[1];
                        }
                    }
                } else if (tl == 0) {
                    // too many windows share the median bins: the generic kernel recomputes the cell
                    const int slot = atomicAdd(P.row_count, 1);
                    P.row_list[slot] = cell - 2 * (int64_t)gridDim.x;
                }
            }
        }
        int2 hv = make_int2(0, 0);
        int htot = 0, hincl = 0, nanf = 0;
        if (have1) {
            // ---- histogram of cell it-1: 4 bins per thread, wavefront prefix sums, clear -------
            nanf = sc->nanflag[p1];
            hv = reinterpret_cast<const int2*>(hist)[tl];
            reinterpret_cast<int2*>(hist)[tl] = make_int2(0, 0);
            const int s = hv.x + hv.y;  // no carry between halves: counts <= W < 65536
            htot = (s & 0xffff) + ((unsigned)s >> 16);
            hincl = wave_scan_dpp(htot);
            if ((tl & 63) == 63) sc->wtot[p1][tl >> 6] = hincl;
        }
        if (have0) {
            // ---- S: block partial sums of cell it, straight into their own LDS region ----------
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int b = tl + i * XT;
                if (b < NB) {
                    double s0 = 0.0, s1 = 0.0;
                    const float* rp = row + b * BT;
                    if constexpr ((BT & 1) == 0) {
                        const float2* rp2 = reinterpret_cast<const float2*>(rp);
                        float2 v2[BT / 2];
#pragma unroll
                        for (int r = 0; r < BT / 2; ++r) v2[r] = rp2[r];
#pragma unroll
                        for (int r = 0; r < BT / 2; ++r) {
                            block_accumulate((double)v2[r].x, 2 * r, s0, s1);
                            block_accumulate((double)v2[r].y, 2 * r + 1, s0, s1);
                        }
                    } else {
                        float v1[BT];
#pragma unroll
                        for (int r = 0; r < BT; ++r) v1[r] = rp[r];
#pragma unroll
                        for (int r = 0; r < BT; ++r) block_accumulate((double)v1[r], r, s0, s1);
                    }
                    *reinterpret_cast<double2*>(S01 + 2 * b) = make_double2(s0, s1);
                }
            }
        }
        ICV_XPH(0)
        __syncthreads();  // barrier 1: {S0,S1} of cell it complete, row dead; scan totals / candidates published
        ICV_XPH(1)
        asm volatile("" : "+v"(tl));

        // =============================== phase B ================================================
        if (have2 && tl < 64) {
            // ---- exact float64 ranks of the <= 64 candidates of cell it-2: one wavefront --------
            const int4 s = *reinterpret_cast<const int4*>(sc->sel[p0]);
            const int2 s2 = *reinterpret_cast<const int2*>(sc->sel[p0] + 4);
            const int nin = s.w + (s.y != s.x ? s2.x : 0);
            double ma = 0.0, mb = 0.0;  // handed-back cell: placeholder, rewritten by k_smooth
            if (s2.y) {
                ma = mb = __builtin_nan("");
            } else if (nin <= 64) {
                const int n = sc->ncand[p0] < 64 ? sc->ncand[p0] : 64;
                const double mine = (tl < n) ? sc->cand[p0][tl] : __builtin_inf();
                int r = 0;
                for (int q = 0; q < n; ++q) {  // n is wavefront-uniform (typically 2..4)
                    const double o = readlane_d(mine, q);
                    r += (int)(o < mine) | ((int)(o == mine) & (int)(q < tl));
                }
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(tl < n && r == k1 - s.z);
                const unsigned long long m2 = __builtin_amdgcn_ballot_w64(tl < n && r == k2 - s.z);
                ma = readlane_d(mine, m1 ? (int)__builtin_ctzll(m1) : 0);
                mb = readlane_d(mine, m2 ? (int)__builtin_ctzll(m2) : 0);
            }
            if (tl == 0) {
                *reinterpret_cast<double2*>(sc->med[p0]) = make_double2(ma, mb);
                sc->ncand[p0] = 0;
            }
        }
        if (have1) {
            // ---- locate the bins of ranks k1, k2 of cell it-1 (the wavefront(s) that hold them) --
            if (nanf) {
                if (tl == 0) {
                    sc->nanflag[p1] = 0;
                    sc->sel[p1][5] = 1;
                }
            } else {
                const int lane = tl & 63;
                const int wv_id = __builtin_amdgcn_readfirstlane(tl >> 6);
                int pre = sc->wtot[p1][lane & 15];
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xf, 0xf, false);
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x112, 0xf, 0xf, false);
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x114, 0xf, 0xf, false);
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x118, 0xf, 0xf, false);
                const int mine = __builtin_amdgcn_readlane(hincl, 63);
                const int base = __builtin_amdgcn_readlane(pre, wv_id) - mine;  // windows in lower wavefronts' bins
                if (tl == 0) sc->sel[p1][5] = 0;
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const int k = which == 0 ? k1 : k2;
                    if (k >= base && k < base + mine) {  // wavefront-uniform
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(hincl + base > k);
                        const int L = (int)__builtin_ctzll(m);
                        const int ex = __builtin_amdgcn_readlane(hincl - htot, L) + base;
                        const int w0 = __builtin_amdgcn_readlane(hv.x, L), w1 = __builtin_amdgcn_readlane(hv.y, L);
                        // lanes 0..3: count of bin i of the located group of four, prefix over the 4 lanes
                        const int word = (lane & 2) == 0 ? w0 : w1;
                        const int cnt = lane < 4 ? ((word >> ((lane & 1) * 16)) & 0xffff) : 0;
                        int inc = cnt;
                        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);
                        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);
                        const unsigned long long mj = __builtin_amdgcn_ballot_w64(lane < 4 && inc + ex > k);
                        const int j = (int)__builtin_ctzll(mj);
                        const int bin = ((wv_id << 6) + L) * 4 + j;
                        const int below = __builtin_amdgcn_readlane(inc - cnt, j) + ex;
                        const int cj = __builtin_amdgcn_readlane(cnt, j);
                        if (lane == 0) {
                            if (which == 0) {
                                sc->sel[p1][0] = bin;
                                sc->sel[p1][2] = below;
                                sc->sel[p1][3] = cj;
                            } else {
                                sc->sel[p1][1] = bin;
                                sc->sel[p1][4] = cj;
                            }
                        }
                    }
                }
            }
        }
        // rotate the window registers: cell it-1 -> wv1, cell it-2 -> wv2
#pragma unroll
        for (int i = 0; i < MAXW; ++i) {
            wv2[i] = wv1[i];
            wv1[i] = wv0[i];
        }
        wb1 = wb0;

        auto w_phase = [&]() __attribute__((always_inline)) {
            int lnan = 0;
            if (wfull) {
                // every window of the wavefront is a full pyramid window: no per-window branches, both windows of
                // the thread advance together (interleaved float64 chains), canonical order inside a window
                constexpr int HB = NBW / 2;
                const double2* sp[MAXW];
                double v[MAXW];
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    sp[i] = reinterpret_cast<const double2*>(S01) + (wdesc[i] & 0xffff);
                    v[i] = 0.0;
                }
#pragma unroll
                for (int m = 0; m < NBW; ++m) {
                    double2 sv[MAXW];
#pragma unroll
                    for (int i = 0; i < MAXW; ++i) sv[i] = sp[i][m];
#pragma unroll
                    for (int i = 0; i < MAXW; ++i)
                        v[i] = fma((double)(m < HB ? m * BT + 1 : NBW * BT - m * BT), sv[i].x, v[i]);
#pragma unroll
                    for (int i = 0; i < MAXW; ++i) v[i] = m < HB ? v[i] + sv[i].y : v[i] - sv[i].y;
                    if (m % ICV_X_WCH == ICV_X_WCH - 1) __builtin_amdgcn_sched_barrier(0);  // bound the reads in flight
                }
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    v[i] = finish_window(v[i], NBW * BT, pyr_den, pyr_rcp, 1.0);
                    const bool valid = tl + i * XT < W;
                    wv0[i] = valid ? v[i] : 0.0;
                    lnan |= valid & (v[i] != v[i]);
                    const int hb = hist_bin(v[i], inv_bound);
                    wb0 = i ? (wb0 | ((unsigned)hb << 16)) : (unsigned)hb;
                    if (valid) atomicAdd(&hist[hb >> 1], 1 << ((hb & 1) * 16));
                }
            } else {
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    const int j = tl + i * XT;
                    wv0[i] = 0.0;
                    if (i == 0) wb0 = 0;
                    if (j < W) {
                        int wp = wdesc[i];
                        asm volatile("" : "+v"(wp));  // decode inside the loop (register budget)
                        const int ln = wp >> 16;
                        const double2* sp = reinterpret_cast<const double2*>(S01) + (wp & 0xffff);
                        double v = window_from_blocks(ln, BT, [&](int m, double& a, double& b2) {
                            const double2 s = sp[m];
                            a = s.x;
                            b2 = s.y;
                        });
                        // flat windows (one per chromosome with <= window genes) read their gene count
                        v = finish_window(v, ln, pyr_den, pyr_rcp, ln > 0 ? 1.0 : P.w_denom[j]);
                        wv0[i] = v;
                        lnan |= (v != v);
                        const int hb = hist_bin(v, inv_bound);
                        wb0 = i ? (wb0 | ((unsigned)hb << 16)) : (unsigned)hb;
                        atomicAdd(&hist[hb >> 1], 1 << ((hb & 1) * 16));
                    }
                }
            }
            if (lnan) sc->nanflag[p0] = 1;  // benign race: every writer stores 1
        };

        const int64_t nxt = cell + gridDim.x;
#if ICV_X_WFIRST == 2
        const bool l_first = (__builtin_amdgcn_readfirstlane(tl >> 6) & 1) != 0;
        if (l_first && nxt < P.n_rows) l_phase(nxt + gridDim.x);
        if (have0) w_phase();
        if (!l_first && nxt < P.n_rows) l_phase(nxt + gridDim.x);
#else
        if (have0) w_phase();
        if (nxt < P.n_rows) l_phase(nxt + gridDim.x);
#endif
        ICV_XPH(2)
        __syncthreads();  // barrier 2: row of cell it+1 scattered, histogram of cell it complete, {S0,S1} dead
        ICV_XPH(3)
    }
#ifdef ICV_X_PROFILE
    if (P.dbg && t == 64)
        for (int i = 0; i < 4; ++i) atomicAdd(P.dbg + i, tacc[i]);
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

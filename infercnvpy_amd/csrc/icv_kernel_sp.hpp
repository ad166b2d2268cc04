// k_smooth_sp: the dense float32 hot path with the workgroup SPLIT into producer and consumer wavefronts.
//
// k_smooth_ws runs two 512-thread workgroups per CU, every wavefront of a workgroup in the same phase; the
// LDS-heavy work of a cell (scatter, {S0,S1} reads, histogram atomics: ~7 700 LDS cycles) and its VALU-heavy
// work (float64 sums) overlap only by chance between the two workgroups, and each cell pays two exposed L2
// round trips for the reference / scatter tables (knock-out and occupancy experiments in DESIGN.md).
// Here ONE 1024-thread workgroup owns the CU:
//
//   wavefronts 0..7  (producers): scatter the prefetched row of cell n into LDS (L), block sums (S)
//   wavefronts 8..15 (consumers): windows (W), histogram, median, x_res and moments, one / two cells behind
//
// Both halves run in lock step through the same four s_barriers per iteration, paired so that an LDS-heavy
// part always runs next to a light or VALU-heavy one:
//
//   seg | producers (cell n = it)                            | consumers
//   ----+-----------------------------------------------------+----------------------------------------------
//   A0  | scatter vectors 0-3; reload them (row n+1, tables)  | moments of cell it-3 to HBM; histogram of cell it-2
//   A1  | scatter vectors 4-6; reload                         | every wavefront scans the whole histogram and
//       |                                                     | locates the two middle bins; candidates -> cand[]
//   A2  | scatter vectors 7-9; reload                         | clear the histogram; exact float64 ranks -> median
//   B   | block sums of row n -> {S0,S1}[n & 1]               | x_res + moments of cell it-2; windows of cell it-1
//       |                                                     | from {S0,S1}[(n-1) & 1]
//
// {S0,S1} is double buffered and neither it nor the histogram aliases the row (154 KB of LDS for the one
// workgroup): padding positions are zeroed once, and every row / table vector is re-requested the moment
// its old value has been scattered, a whole iteration before it is needed -- no HBM or L2 latency is exposed.
//
// All float64 arithmetic is the canonical sequence of icv_kernels.hpp (bit-identical to k_smooth).
//
// STATUS (round 1): opt-in (ICV_SP=1), parity-tested, NOT the default -- 2.7 ms per 100 000 cells against 2.1 ms
// for k_smooth_ws.  With one workgroup per CU only one row (80 KB) is in flight per CU, and the knock-out runs
// (everything but the loads, barriers and stores removed: still 1.94 ms) show that the path is bound by memory
// level parallelism -- bytes in flight x ~5 us loaded HBM latency -- before any LDS / VALU balance matters;
// see DESIGN.md section 4 "what binds the smoothing kernel".  A partial two-rows-ahead prefetch (the first 2 / 4 of
// the 10 row vectors requested one iteration earlier, +20 / +40 % bytes in flight) was measured afterwards and
// changed nothing (2.54 ms; the 4-vector version spills and is slower), so the floor is not the number of rows in
// flight alone: the four 16-wavefront barriers per cell, the consumers' serial chain (about 14 k cycles per cell
// against 5 k for the producers) and the texture path are the open questions for the next round.
#pragma once
#include "icv_kernel_ws.hpp"

namespace icv {

#ifndef ICV_SP_WGROUP
#define ICV_SP_WGROUP 2
#endif
constexpr int SPT = 1024;  // threads per workgroup: 8 producer + 8 consumer wavefronts
constexpr int SPH = 512;   // threads per half

template <int UMAX, int MAXB, int BT, int NBW>
__global__ void __launch_bounds__(SPT) k_smooth_sp(const KParams P) {
    static_assert(NBIN == 8 * SPH, "the histogram scan gives every consumer thread 8 bins");
    static_assert(MAXB == 4, "four block-sum segments per iteration");
    constexpr int MAXW = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* row = reinterpret_cast<float*>(smem);
    double* S01 = reinterpret_cast<double*>(smem + P.win_off);
    int* hist = reinterpret_cast<int*>(smem + P.hist_off);
    ScratchW* sc = reinterpret_cast<ScratchW*>(smem + P.scratch_off);

    const int t = threadIdx.x;
    const bool producer = __builtin_amdgcn_readfirstlane(t >> 6) < 8;  // wave-uniform
    const int tg = t & (SPH - 1);                                       // index inside the half
    const int W = P.W, NB = P.NB;
    const int k1 = (W - 1) / 2, k2 = W / 2;
    const float inv_bound = (float)(1.0 / P.med_bound);
    const float cap = (float)P.cap;
    const unsigned row_bytes = (unsigned)P.n_cols * 4u;
    const unsigned voff = (unsigned)tg * 16u, voff8 = (unsigned)tg * 8u;
    const double pyr_den = P.pyr_den, pyr_rcp = P.pyr_rcp;
    const float* xbase = static_cast<const float*>(P.values);
    const int64_t n_mine = (P.n_rows - blockIdx.x + gridDim.x - 1) / gridDim.x;

    // ---- one-time initialisation: padding positions, histogram, scratch ----------------------------
    for (int i = t; i < P.n_pad; i += SPT) row[P.pad_idx[i]] = 0.0f;
    for (int i = t; i < NBIN / 2; i += SPT) hist[i] = 0;
    if (t == 0) {
        sc->nanflag = 0;
        sc->mode = 1;
        sc->ncand = 0;
        sc->ma = 0.0;
        sc->mb = 0.0;
    }

    // developer diagnostic (-DICV_SP_PROFILE, ICV_PHASE_PROFILE=1): work and barrier-wait cycles per segment
#ifdef ICV_SP_PROFILE
    unsigned long long pt_last = __builtin_amdgcn_s_memtime(), pt_work[5] = {0, 0, 0, 0, 0}, pt_wait[5] = {0, 0, 0, 0, 0};
#define SP_BAR(seg)                                                        \
    {                                                                      \
        unsigned long long a_ = __builtin_amdgcn_s_memtime();              \
        __syncthreads();                                                   \
        unsigned long long b_ = __builtin_amdgcn_s_memtime();              \
        pt_work[seg] += a_ - pt_last;                                      \
        pt_wait[seg] += b_ - a_;                                           \
        pt_last = b_;                                                      \
    }
#else
#define SP_BAR(seg) __syncthreads();
#endif
    if (producer) {
        // =========================== producers ===============================================
        const __amdgpu_buffer_rsrc_t lo_rs = make_rsrc(P.ref_lo, row_bytes);
        const __amdgpu_buffer_rsrc_t d16_rs = make_rsrc(P.dst16, (unsigned)(UMAX * SPH * 8));
        u32x4 xq[UMAX], lo[UMAX];
        u32x2 dd[UMAX];
        {
            const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + (int64_t)blockIdx.x * P.ld, row_bytes);
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {
                xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * SPH * 16, 0);
                lo[u] = __builtin_amdgcn_raw_buffer_load_b128(lo_rs, voff, u * SPH * 16, 0);
                dd[u] = __builtin_amdgcn_raw_buffer_load_b64(d16_rs, voff8, u * SPH * 8, 0);
            }
        }
        __syncthreads();  // initialisation visible
        for (int64_t it = 0; it < n_mine + 2; ++it) {
            const bool mine = it < n_mine;
            // the row after this one (clamped: the reload is unconditional so that no register stays live
            // around the loop; the last reloads fetch a valid row again and are never used)
            const int64_t nit = it + 1 < n_mine ? it + 1 : (n_mine > 0 ? n_mine - 1 : 0);
            const __amdgpu_buffer_rsrc_t xr =
                make_rsrc(xbase + ((int64_t)blockIdx.x + nit * gridDim.x) * P.ld, row_bytes);
            int tl = tg;
            asm volatile("" : "+v"(tl));  // keep thread-derived addresses out of LICM (register budget)
            // ---- seg A0..A3: scatter, a few vectors per segment -----------------------------------
#define SP_SCATTER(U0, U1)                                                                                    \
    {                                                                                                         \
        bool any_nan = false;                                                                                 \
        _Pragma("unroll") for (int u = U0; u < U1; ++u) {                                                     \
            const float y0 = __uint_as_float(xq[u].x) - __uint_as_float(lo[u].x);                             \
            const float y1 = __uint_as_float(xq[u].y) - __uint_as_float(lo[u].y);                             \
            const float y2 = __uint_as_float(xq[u].z) - __uint_as_float(lo[u].z);                             \
            const float y3 = __uint_as_float(xq[u].w) - __uint_as_float(lo[u].w);                             \
            any_nan |= __builtin_isunordered(y0, y1) | __builtin_isunordered(y2, y3);                         \
            if (mine) {                                                                                       \
                row[dd[u].x & 0xffffu] = __builtin_amdgcn_fmed3f(y0, -cap, cap);                              \
                row[dd[u].x >> 16] = __builtin_amdgcn_fmed3f(y1, -cap, cap);                                  \
                row[dd[u].y & 0xffffu] = __builtin_amdgcn_fmed3f(y2, -cap, cap);                              \
                row[dd[u].y >> 16] = __builtin_amdgcn_fmed3f(y3, -cap, cap);                                  \
            }                                                                                                 \
        }                                                                                                     \
        if (__builtin_expect(any_nan && mine, 0)) { /* v_med3 drops NaNs, np.clip keeps them */              \
            _Pragma("unroll") for (int u = U0; u < U1; ++u) {                                                 \
                const float y0 = __uint_as_float(xq[u].x) - __uint_as_float(lo[u].x);                         \
                const float y1 = __uint_as_float(xq[u].y) - __uint_as_float(lo[u].y);                         \
                const float y2 = __uint_as_float(xq[u].z) - __uint_as_float(lo[u].z);                         \
                const float y3 = __uint_as_float(xq[u].w) - __uint_as_float(lo[u].w);                         \
                if (y0 != y0) row[dd[u].x & 0xffffu] = y0;                                                    \
                if (y1 != y1) row[dd[u].x >> 16] = y1;                                                        \
                if (y2 != y2) row[dd[u].y & 0xffffu] = y2;                                                    \
                if (y3 != y3) row[dd[u].y >> 16] = y3;                                                        \
            }                                                                                                 \
        }                                                                                                     \
        _Pragma("unroll") for (int u = U0; u < U1; ++u) {                                                     \
            xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * SPH * 16, 0);                         \
            lo[u] = __builtin_amdgcn_raw_buffer_load_b128(lo_rs, voff, u * SPH * 16, 0);                      \
            dd[u] = __builtin_amdgcn_raw_buffer_load_b64(d16_rs, voff8, u * SPH * 8, 0);                      \
        }                                                                                                     \
    }
            SP_SCATTER(0, 4)
            SP_BAR(0)
            SP_SCATTER(4, 7)
            SP_BAR(1)
            SP_SCATTER(7, UMAX)
            SP_BAR(2)  // row n complete
#undef SP_SCATTER
            // ---- seg B: block sums -> {S0,S1}[it & 1] ------------------------------------------------
            if (mine) {
                double* S01w = S01 + (size_t)(it & 1) * 2 * NB;
#pragma unroll
                for (int i = 0; i < MAXB; ++i) {
                    const int b = tl + i * SPH;
                    if (b < NB) {
                        double s0 = 0.0, s1 = 0.0;
                        const float* rp = row + b * BT;
                        if constexpr ((BT & 1) == 0) {
                            const float2* rp2 = reinterpret_cast<const float2*>(rp);
#pragma unroll
                            for (int r = 0; r < BT; r += 2) {
                                const float2 v2 = rp2[r >> 1];
                                block_accumulate((double)v2.x, r, s0, s1);
                                block_accumulate((double)v2.y, r + 1, s0, s1);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < BT; ++r) block_accumulate((double)rp[r], r, s0, s1);
                        }
                        *reinterpret_cast<double2*>(S01w + 2 * b) = make_double2(s0, s1);
                    }
                }
            }
            SP_BAR(3)
        }
#ifdef ICV_SP_PROFILE
        if (P.dbg && t == 0)
            for (int i = 0; i < 5; ++i) {
                atomicAdd(P.dbg + i, pt_work[i]);
                atomicAdd(P.dbg + 5 + i, pt_wait[i]);
            }
#endif
    } else {
        // =========================== consumers ===============================================
        const __amdgpu_buffer_rsrc_t wp_rs = make_rsrc(P.w_pack, (unsigned)W * 4u);
        int w_pack[MAXW];
#pragma unroll
        for (int i = 0; i < MAXW; ++i)  // out-of-range windows read 0 (buffer bounds check)
            w_pack[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(wp_rs, (unsigned)tg * 4u, i * SPH * 4, 0);
        double wv[MAXW];
        unsigned wbin[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < MAXW; ++i) wv[i] = 0.0;
        __syncthreads();  // initialisation visible
        for (int64_t it = 0; it < n_mine + 2; ++it) {
            const bool have_med = it >= 2;                     // cell it-2: its median is found in A0..A2
            const bool have_win = it >= 1 && it - 1 < n_mine;  // cell it-1: windows in seg B
            const int64_t mcell = (int64_t)blockIdx.x + (it - 2) * gridDim.x;
            int tl = tg;
            asm volatile("" : "+v"(tl));
            // ---- seg A0: moments of cell it-3 to HBM; histogram of cell it-2 --------------------------------
#ifdef ICV_SP_PROFILE
            const unsigned long long f0_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef ICV_SP_PROFILE
            const unsigned long long f1_ = __builtin_amdgcn_s_memtime();
            if (P.dbg && t == SPH) atomicAdd(P.dbg + 20, f1_ - f0_);
#endif
            if (have_med) {
                int lnan = 0;
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    if (tl + i * SPH < W) {
                        const double v = wv[i];
                        lnan |= (v != v);
                        const int hb = hist_bin(v, inv_bound);
                        wbin[i >> 1] = (i & 1) ? (wbin[i >> 1] | ((unsigned)hb << 16)) : (unsigned)hb;
                        atomicAdd(&hist[hb >> 1], 1 << ((hb & 1) * 16));  // 16-bit bins, two per word
                    }
                }
                if (lnan) sc->nanflag = 1;  // benign race: every writer stores 1
                if (tl == 0) sc->ncand = 0;
            }
#ifdef ICV_SP_PROFILE
            {
                const unsigned long long f2_ = __builtin_amdgcn_s_memtime();
                if (P.dbg && t == SPH) atomicAdd(P.dbg + 21, f2_ - f1_);
            }
#endif
            SP_BAR(0)
            // ---- seg A1: EVERY consumer wavefront scans the whole histogram (no publish step, no barrier
            //      between scan and gather), then the windows in the two median bins go to cand[] ----------
            int nanf = 0, mode = 1, below = 0;
            if (have_med) {
                nanf = sc->nanflag;
                if (!nanf) {
                    const int lane = tl & 63;
                    // level 1: lane l sums bins [64 l, 64 l + 64) (two 16-bit bins per word)
                    const int4* h4 = reinterpret_cast<const int4*>(hist) + lane * 8;
                    int tot = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int4 v = h4[q];
                        const int s4 = (v.x + v.y) + (v.z + v.w);  // no carry: counts <= W < 65536
                        tot += (s4 & 0xffff) + ((unsigned)s4 >> 16);
                    }
                    const int incl = wave_scan_dpp(tot);
                    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(incl > k1);
                    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(incl > k2);
                    const int l1 = m1 ? (int)__builtin_ctzll(m1) : 63, l2 = m2 ? (int)__builtin_ctzll(m2) : 63;
                    const int ex1 = __builtin_amdgcn_readlane(incl - tot, l1);
                    const int ex2 = __builtin_amdgcn_readlane(incl - tot, l2);
                    // level 2: lane i looks at bin 64 l + i of the located group
                    const int c1 = (hist[l1 * 32 + (lane >> 1)] >> ((lane & 1) * 16)) & 0xffff;
                    const int c2 = (hist[l2 * 32 + (lane >> 1)] >> ((lane & 1) * 16)) & 0xffff;
                    const int in1 = wave_scan_dpp(c1) + ex1, in2 = wave_scan_dpp(c2) + ex2;
                    const unsigned long long n1 = __builtin_amdgcn_ballot_w64(in1 > k1);
                    const unsigned long long n2 = __builtin_amdgcn_ballot_w64(in2 > k2);
                    const int j1 = n1 ? (int)__builtin_ctzll(n1) : 63, j2 = n2 ? (int)__builtin_ctzll(n2) : 63;
                    const int b1 = l1 * 64 + j1, b2 = l2 * 64 + j2;
                    below = __builtin_amdgcn_readlane(in1 - c1, j1);
                    const int n_in_bins =
                        __builtin_amdgcn_readlane(c1, j1) + (b2 != b1 ? __builtin_amdgcn_readlane(c2, j2) : 0);
                    if (n_in_bins <= 64) {
                        mode = 0;
#pragma unroll
                        for (int i = 0; i < MAXW; ++i) {
                            if (tl + i * SPH < W) {
                                const int b = (int)((wbin[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
                                if (b == b1 || b == b2) {
                                    const int idx = atomicAdd(&sc->ncand, 1);
                                    if (idx < 64) sc->cand[idx] = wv[i];
                                }
                            }
                        }
                    } else {
                        mode = 2;  // too many windows share the median bins: k_smooth recomputes the cell
                        if (tl == 0) {
                            const int slot = atomicAdd(P.row_count, 1);
                            P.row_list[slot] = mcell;
                        }
                    }
                }
            }
            SP_BAR(1)
            // ---- seg A2: clear the histogram; exact float64 ranks of the <= 64 candidates ---------------------
            if (have_med) {
                reinterpret_cast<int4*>(hist)[tl] = make_int4(0, 0, 0, 0);
                if (nanf && tl == 0) sc->nanflag = 0;
                if (mode == 0) {
                    const int n = sc->ncand < 64 ? sc->ncand : 64;
                    const int ci = tl >> 3, part = tl & 7;
                    const double mine_raw = sc->cand[ci];
                    const double2* cp = reinterpret_cast<const double2*>(sc->cand + part * 8);
                    const double2 o01 = cp[0], o23 = cp[1], o45 = cp[2], o67 = cp[3];
                    const double o[8] = {o01.x, o01.y, o23.x, o23.y, o45.x, o45.y, o67.x, o67.y};
                    const double mine = (ci < n) ? mine_raw : __builtin_inf();
                    int r = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int jj = part * 8 + q;
                        const int hit = (int)(o[q] < mine) | ((int)(o[q] == mine) & (int)(jj < ci));
                        r += hit & (int)(jj < n);
                    }
                    r += dpp_move_i<0xB1>(r);   // lanes ^1
                    r += dpp_move_i<0x4E>(r);   // lanes ^2
                    r += dpp_move_i<0x141>(r);  // row_half_mirror: the other quad of the 8-lane group
                    if (part == 0 && ci < n) {
                        if (r == k1 - below) sc->ma = mine;
                        if (r == k2 - below) sc->mb = mine;
                    }
                }
            }
            SP_BAR(2)
            // ---- seg B: x_res + moments of cell it-2, then the windows of cell it-1 ---------------------------
            if (have_med) {
                double med = (k1 == k2) ? sc->ma : (sc->ma + sc->mb) / 2.0;
                if (mode == 1) med = __builtin_nan("");  // a NaN window: the whole cell is NaN
                if (mode == 2) med = 0.0;
                double sum = 0.0, sq = 0.0;
                float* orow = P.out + mcell * P.ldo;
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    const int j = tl + i * SPH;
                    if (j < W) {
                        const double y = wv[i] - med;
                        orow[j] = (float)y;
                        sum = sum + y;
                        sq = fma(y, y, sq);
                    }
                }
                sum = wave_sum_dpp(sum);
                sq = wave_sum_dpp(sq);
                if ((tl & 63) == 0)  // this wavefront's share of the cell's moments, straight to HBM
                    reinterpret_cast<double2*>(P.cell_part)[mcell * NWAVE + (tl >> 6)] = make_double2(sum, sq);
                if (tl == 0) P.cell_median[mcell] = med;
            }
            if (have_win) {
                const double2* S2 = reinterpret_cast<const double2*>(S01 + (size_t)((it - 1) & 1) * 2 * NB);
                // The consumer half has registers to spare (no row prefetch): the four windows of a thread are
                // evaluated together, 20 ds_read_b128 in flight and four independent float64 chains, instead
                // of one latency-bound chain after the other.  Each window still sees its blocks in ascending
                // order (canonical sequence).  Windows past the end are computed on block 0 and discarded.
                bool regular = true;
                int start[MAXW];
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    const bool valid = tl + i * SPH < W;
                    regular = regular && (!valid || (w_pack[i] >> 16) == NBW * BT);
                    start[i] = valid ? (w_pack[i] & 0xffff) : 0;
                }
                if (regular) {
                    constexpr int HB = NBW / 2;
                    constexpr int WG = ICV_SP_WGROUP;  // windows evaluated together (register budget)
                    double v[MAXW] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int i0 = 0; i0 < MAXW; i0 += WG) {
                        {
                            double2 sb[WG][HB];
#pragma unroll
                            for (int i = 0; i < WG; ++i)
#pragma unroll
                                for (int m = 0; m < HB; ++m) sb[i][m] = S2[start[i0 + i] + m];
#pragma unroll
                            for (int m = 0; m < HB; ++m)
#pragma unroll
                                for (int i = 0; i < WG; ++i) {
                                    v[i0 + i] = fma((double)(m * BT + 1), sb[i][m].x, v[i0 + i]);
                                    v[i0 + i] = v[i0 + i] + sb[i][m].y;
                                }
                        }
                        // the next batch of LDS reads must not be hoisted above this one (register budget): its
                        // addresses are made to depend on the sums just computed
#pragma unroll
                        for (int i = 0; i < WG; ++i) asm volatile("" : "+v"(start[i0 + i]) : "v"(v[i0 + i]));
                        {
                            double2 sb[WG][NBW - HB];
#pragma unroll
                            for (int i = 0; i < WG; ++i)
#pragma unroll
                                for (int m = 0; m < NBW - HB; ++m) sb[i][m] = S2[start[i0 + i] + HB + m];
#pragma unroll
                            for (int m = 0; m < NBW - HB; ++m)
#pragma unroll
                                for (int i = 0; i < WG; ++i) {
                                    v[i0 + i] = fma((double)(NBW * BT - (HB + m) * BT), sb[i][m].x, v[i0 + i]);
                                    v[i0 + i] = v[i0 + i] - sb[i][m].y;
                                }
                        }
                        if (i0 + WG < MAXW) {
#pragma unroll
                            for (int i = 0; i < WG; ++i) asm volatile("" : "+v"(start[i0 + WG + i]) : "v"(v[i0 + i]));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < MAXW; ++i)
                        wv[i] = (tl + i * SPH < W) ? finish_window(v[i], NBW * BT, pyr_den, pyr_rcp, 1.0) : 0.0;
                } else {
                    // a flat window (one per chromosome with <= window genes) among the four: per-window path;
                    // its gene count comes from the window table (a rare global load)
#pragma unroll 1
                    for (int i = 0; i < MAXW; ++i) {
                        const int j = tl + i * SPH;
                        double v = 0.0;
                        if (j < W) {
                            const int wp = i == 0 ? w_pack[0] : (i == 1 ? w_pack[1] : (i == 2 ? w_pack[2] : w_pack[3]));
                            const int ln = wp >> 16;
                            const double2* sp = S2 + (wp & 0xffff);
                            v = window_from_blocks(ln, BT, [&](int m, double& a2, double& b2) {
                                const double2 sv = sp[m];
                                a2 = sv.x;
                                b2 = sv.y;
                            });
                            v = finish_window(v, ln, pyr_den, pyr_rcp, ln > 0 ? 1.0 : P.w_denom[j]);
                        }
                        if (i == 0) wv[0] = v;
                        else if (i == 1) wv[1] = v;
                        else if (i == 2) wv[2] = v;
                        else wv[3] = v;
                    }
                }
            }
            SP_BAR(3)
        }
#ifdef ICV_SP_PROFILE
        if (P.dbg && t == SPH)
            for (int i = 0; i < 5; ++i) {
                atomicAdd(P.dbg + 10 + i, pt_work[i]);
                atomicAdd(P.dbg + 15 + i, pt_wait[i]);
            }
#endif
    }
#undef SP_BAR
}

}  // namespace icv

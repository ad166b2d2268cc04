// k_smooth_sd: CSR input with long windows -- the work of a cell is proportional to its stored entries.
//
// A cell without stored entries is the pre-centred zero row z = clip(0 - ref): its window sums ("base", k_sd_base)
// are the same for every cell.  A stored entry changes one gene from z[g] to v = clip(x - ref[g]); windows are linear
// in the gene values, so   window(cell) = (base + sum over the stored entries of weight * (v - z[g])) / denominator.
// The kernel never builds the 20 000-gene row: it reads a cell's stored entries {column, value} as they are, fetches
// the column's table entry {LDS slot of its block | offset inside the block, reference, z} (k_sd_table: 16 bytes per
// gene, L2 resident) and adds the differences d = v - z[g] into per-block bins {S0 = sum d, S1 = sum j d} (j: gene
// offset inside the block).  The bins become prefix sums over the blocks, and a pyramid window is a linear
// combination of three of them (see phase 4); a flat window a difference of two.
//
// The bins are 64-bit fixed point (d * 2^k, k from the clip value: sd_fraction_bits in icv_api.hip), added with LDS
// integer atomics: integer addition is associative, so the result depends neither on the order of the entries in the
// row nor on the order in which the atomics land (bit-reproducible), and the quantisation (2^-k <= 2^-46 per entry,
// 2^-48 at the default clip of 3) is far below the float64 rounding of the prefix differences.  Windows agree with the
// canonical evaluation order of k_smooth to ~1e-12.
//
// Per cell and workgroup (512 threads, two workgroups per CU), the median of cell k-1 shares the barriers of cell k:
//   phase 0   histogram of k-1: 8 bins per thread, wavefront prefix sums      | bins of k zeroed; table entries of the
//                                                                               columns of k requested
//   barrier A
//   phase 1   middle bins of k-1 located, histogram cleared                   | entries of k added to the bins
//   barrier B1
//   phase 2   windows of k-1 in the middle bins gathered (<= 64)              | bins -> float64 prefix sums per wavefront
//   barrier B2
//   phase 3   candidates ranked exactly -> median of k-1                      | sums of the wavefronts before each one;
//             window tables and {column, value} of k+1 requested
//   barrier B3
//   phase 4   x_res, moments of k-1 stored                                    | windows of k, histogram of k
//   barrier B4
// The kernel is bound by VALU issue (16 wavefronts per CU: LDS holds two workgroups), so work is skipped by branches
// wherever a wavefront has none -- a branch-free window path and a version with four barriers and tail arrays instead
// of the wavefront offsets both executed more instructions and were slower (10.3 and 10.5 ms against 8.8).
// A NaN among the stored values of a cell (never on real data) and a cell with more than 64 windows in its median
// bins are handed back to the generic k_smooth (row_list), like k_smooth_ws does.
#pragma once
#include "icv_kernel_ws.hpp"

// Wave priority by phase (s_setprio): the SIMD's arbiter prefers the wavefronts of the workgroup that is in its
// scan (phase 2) or window (phase 4) phase over the one that zeroes bins, waits for gathers or issues loads:
// 8.40 -> 7.90 ms on config 4 (same box; priority 2 in phases 1, 2, 4: 8.09; in phases 0 and 3: 8.43; 3 instead of 2
// in phases 2 and 4: 7.82, within the noise)
#ifndef ICV_SD_P0
#define ICV_SD_P0 0
#define ICV_SD_P1 0
#define ICV_SD_P2 2
#define ICV_SD_P3 0
#define ICV_SD_P4 2
#endif
#define ICV_SD_PRIO(p) \
    if (ICV_SD_P0 | ICV_SD_P1 | ICV_SD_P2 | ICV_SD_P3 | ICV_SD_P4) __builtin_amdgcn_s_setprio(p);

namespace icv {

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_shift0(double v) {  // lanes without a source receive 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
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
// 16-byte pairs, plane stride 513 pairs): the eight accesses of a thread that owns blocks 8t .. 8t+7 (one plane,
// consecutive slots over the lanes) are conflict-free
constexpr int kWsPlane = 513;
__device__ __forceinline__ int ws_pidx(int b) { return (b & 7) * kWsPlane + (b >> 3); }

constexpr int kSdPF = 4;  // stored entries prefetched per thread (rows with <= 2048 entries; longer rows fetch the
                          // rest inside phase 1)
constexpr int kSdPlaneBytes = 16 * 8 * kWsPlane;  // bins / prefix sums: 8 planes of 513 pairs (ws_pidx)
constexpr int kSdLds = kSdPlaneBytes + NBIN * 2 + 1536;

struct ScratchD {
    int nanflag;
    int mode;               // 0: gather candidates of bins b1/b2, 1: median final, 2: cell handed back
    int b1, b2;
    int ncand;
    int below;              // windows in bins below b1
    int c1, c2;             // windows in bin b1 / b2
    int wtot[NWAVE];        // histogram scan: windows in the 512 bins scanned by each wavefront
    int handback[2];        // a stored NaN in the cell of an even / odd iteration
    double ma, mb;          // the two middle order statistics of the previous cell
    double psum[NWAVE], psq[NWAVE];  // block prefix sums: wavefront totals of {S0, T1}
    double2 poff[NWAVE];    // ... and the sums of the wavefronts before each one
    double cand[64];
    unsigned long long tacc[14];  // -DICV_SD_PROFILE: shader cycles per phase / barrier wait (thread 64)
};
static_assert(sizeof(ScratchD) <= 1536, "ScratchD must fit the scratch region");

// per input column: {LDS slot of the block's bins | gene offset inside the block << 16 (all ones: masked column), ref_lo, ref_hi,
// z = clip(0 - ref)} -- everything phase 1 needs to turn a stored value into its difference to the zero row
__global__ void __launch_bounds__(256) k_sd_table(const KParams P, u32x4* tab) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P.n_cols) return;
    const float* lo = static_cast<const float*>(P.ref_lo);
    const float* hi = P.bounded ? static_cast<const float*>(P.ref_hi) : lo;
    const int pos = P.dst[g];
    u32x4 e = {0xffffffffu, 0u, 0u, 0u};
    if (pos >= 0) {
        const int blk = pos / P.B;  // its bins: 64-bit words 2 ws_pidx(blk), + 1
        e.x = (uint32_t)(2 * ws_pidx(blk)) | ((uint32_t)(pos - blk * P.B) << 16);
        e.y = __float_as_uint(lo[g]);
        e.z = __float_as_uint(hi[g]);
        e.w = __float_as_uint(centre_clip<float>(0.0f, lo[g], hi[g], (float)P.cap, P.bounded, P.trunc));
    }
    tab[g] = e;
}

// window sums of the zero row (numerators: before the division by the pyramid weight sum / the gene count)
__global__ void __launch_bounds__(256) k_sd_base(const KParams P, const float* zrow, double* base) {
    const int j = blockIdx.x * 256 + threadIdx.x;
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
    base[j] = acc;
}

template <int MAXW>
__global__ void __launch_bounds__(NT, 4) k_smooth_sd(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2* SP = reinterpret_cast<double2*>(smem);
    unsigned long long* SQ = reinterpret_cast<unsigned long long*>(smem);
    int* hist = reinterpret_cast<int*>(smem + kSdPlaneBytes);
    ScratchD* sc = reinterpret_cast<ScratchD*>(smem + kSdPlaneBytes + NBIN * 2);

    const int t = threadIdx.x;
    static_assert(NBIN == 8 * NT, "the histogram scan gives every thread 8 bins");
    const int W = P.W, NB = P.NB, B = P.B;
    const int win = P.sd_window, nbw = win / B, hbw = nbw / 2;
    const int k1 = (W - 1) / 2, k2 = W / 2;
    const float inv_bound = (float)(1.0 / P.med_bound);
    const double pyr_den = P.pyr_den, pyr_rcp = P.pyr_rcp, qinv = P.sd_qinv;
    const int64_t n_mine = (P.n_rows - blockIdx.x + gridDim.x - 1) / gridDim.x;

    if (t == 0) {
        sc->nanflag = 0;
        sc->mode = 1;
        sc->ncand = 0;
        sc->handback[0] = 0;
        sc->handback[1] = 0;
        sc->ma = 0.0;
        sc->mb = 0.0;
    }
    if (t < NWAVE) {  // wavefronts whose blocks 512 w .. 512 w + 511 lie past the last block never publish totals
        sc->psum[t] = 0.0;
        sc->psq[t] = 0.0;
    }
    // this wavefront owns blocks at all (wavefront-uniform): geometries with few blocks (window 100: 2 000) skip the
    // zeroing and the scan of the planes' unused tail
    const bool wblk = ((t & ~63) * 8) < NB;
    // per-thread tables, re-read from L2 in every cell (as loop-carried registers they end up in scratch): window
    // descriptor, gene offset and zero-row sum of windows t, t + 512, ...; first-gene offsets of blocks 8 t .. 8 t + 7
    const __amdgpu_buffer_rsrc_t wp_rs = make_rsrc(P.w_pack, (unsigned)W * 4u);
    const __amdgpu_buffer_rsrc_t sr_rs = make_rsrc(P.w_srel, (unsigned)W * 4u);
    const __amdgpu_buffer_rsrc_t wb_rs = make_rsrc(P.sd_base, (unsigned)W * 8u);
    const __amdgpu_buffer_rsrc_t g0_rs = make_rsrc(P.blk_g0, (unsigned)(NB + 8) * 4u);
    // row offsets travel through vector loads (an address that the compiler can prove uniform becomes a scalar load,
    // whose counter is shared with the LDS operations of the phase that follows)
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const __amdgpu_buffer_rsrc_t tab_rs = make_rsrc(P.sd_tab, (unsigned)P.n_cols * 16u);
    const float cap = (float)P.cap;
    const double scale = P.sd_scale;
    // Software pipeline over the cells of this workgroup, filled by iteration -1 (which does nothing else: with no
    // loads pending at the loop entry the compiler's wait counts inside the loop are those of the steady state):
    // phase 3 of iteration k requests {column, value} of cell k+1, phase 0 of iteration k+1 the table entries of those
    // columns (the gathers: one L1 line each, kept apart from the loads of phase 3), phase 1 consumes them.
    int64_t a0 = 0;  // first stored entry of the cell of phase 1, and its entry count
    int c0 = 0;
    u32x4 etab[kSdPF];     // table entries of its columns
    unsigned nidx[kSdPF];  // its columns and values
    float nval[kSdPF];
#pragma unroll
    for (int i = 0; i < kSdPF; ++i) {
        nidx[i] = 0u;
        nval[i] = 0.0f;
    }
    double wv[MAXW];  // this thread's windows of the previous cell (x_res needs its median)
    unsigned wbin[(MAXW + 1) / 2];  // their histogram bins, two 16-bit bins per register
#pragma unroll
    for (int i = 0; i < MAXW; ++i) wv[i] = 0.0;
#pragma unroll
    for (int i = 0; i < (MAXW + 1) / 2; ++i) wbin[i] = 0;
    // phase timers (developer build, -DICV_SD_PROFILE): thread 64 accumulates s_memtime deltas in LDS
#ifdef ICV_SD_PROFILE
    unsigned long long tlast = 0;
    if (t == 64)
        for (int i = 0; i < 14; ++i) sc->tacc[i] = 0;
#define ICV_SDP(i)                                              \
    if (P.dbg && t == 64) {                                     \
        unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        sc->tacc[i] += now_ - tlast;                            \
        tlast = now_;                                           \
    }
#else
#define ICV_SDP(i)
#endif
    __syncthreads();
#ifdef ICV_SD_PROFILE
    if (P.dbg && t == 64) tlast = __builtin_amdgcn_s_memtime();
#endif

    for (int64_t it = -1; it <= n_mine; ++it) {
        const bool more = it >= 0 && it < n_mine;  // a cell to smooth in this iteration
        const bool have_prev = it > 0;  // a cell whose median is being resolved
        const int64_t cell = (int64_t)blockIdx.x + it * gridDim.x;
        const int64_t pcell = cell - gridDim.x;
        const int par = (int)(it & 1);
        int tl = t;
        asm volatile("" : "+v"(tl));  // keep thread-derived values out of LICM (see k_smooth_fast)

        // ---------------- phase 0: histogram level 1 (previous cell); bins zeroed (this cell) -------------------
        ICV_SD_PRIO(ICV_SD_P0)
        int4 hv = make_int4(0, 0, 0, 0);
        int htot = 0, hincl = 0, nanf = 0, hback = 0;
        if (have_prev) {
            nanf = sc->nanflag;
            hback = sc->handback[par ^ 1];
            hv = reinterpret_cast<const int4*>(hist)[tl];
            const int s4 = (hv.x + hv.y) + (hv.z + hv.w);  // no carry between halves: counts <= W < 65536
            htot = (s4 & 0xffff) + ((unsigned)s4 >> 16);
            hincl = wave_scan_dpp(htot);
            if ((tl & 63) == 63) sc->wtot[tl >> 6] = hincl;
        }
        const int64_t nxt = cell + gridDim.x;
#pragma unroll
        for (int i = 0; i < kSdPF; ++i) etab[i] = __builtin_amdgcn_raw_buffer_load_b128(tab_rs, nidx[i] * 16u, 0, 0);
        if (more && wblk) {
#pragma unroll
            for (int k = 0; k < 8; ++k) reinterpret_cast<int4*>(SP)[k * kWsPlane + tl] = make_int4(0, 0, 0, 0);
        }
        ICV_SDP(0)
        __syncthreads();  // A: histogram consumed, bins zero; wavefront totals published
        ICV_SDP(1)

        // ---------------- phase 1: middle bins located (previous cell); entries added (this cell) ---------------
        ICV_SD_PRIO(ICV_SD_P1)
        if (have_prev) {
            if (hback) {  // a stored NaN: the generic kernel recomputes the cell
                if (tl == 0) {
                    const int slot = atomicAdd(P.row_count, 1);
                    P.row_list[slot] = pcell;
                    sc->handback[par ^ 1] = 0;
                    sc->nanflag = 0;
                    sc->ma = 0.0;
                    sc->mb = 0.0;
                    sc->mode = 2;
                }
            } else if (nanf) {
                if (tl == 0) {
                    sc->nanflag = 0;
                    sc->ma = __builtin_nan("");
                    sc->mb = __builtin_nan("");
                    sc->mode = 1;
                }
            } else {
                if (tl == 0) {
                    sc->mode = 0;
                    sc->ncand = 0;
                }
                const int wv_id = __builtin_amdgcn_readfirstlane(tl >> 6);
                int base = 0;
#pragma unroll
                for (int u = 0; u < NWAVE; ++u) base += (u < wv_id) ? sc->wtot[u] : 0;
                base = __builtin_amdgcn_readfirstlane(base);
                const int mine = __builtin_amdgcn_readlane(hincl, 63);
                const int lane = tl & 63;
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const int k = which == 0 ? k1 : k2;
                    if (k >= base && k < base + mine) {  // wave-uniform: this wavefront holds rank k
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(hincl + base > k);
                        const int L = (int)__builtin_ctzll(m);
                        const int ex = __builtin_amdgcn_readlane(hincl - htot, L) + base;  // windows below lane L's bins
                        const int w0 = __builtin_amdgcn_readlane(hv.x, L), w1 = __builtin_amdgcn_readlane(hv.y, L);
                        const int w2 = __builtin_amdgcn_readlane(hv.z, L), w3 = __builtin_amdgcn_readlane(hv.w, L);
                        // lanes 0..7: count of bin i of the located group, prefix over 8 lanes (one DPP row)
                        const int word = (lane & 6) == 0 ? w0 : ((lane & 6) == 2 ? w1 : ((lane & 6) == 4 ? w2 : w3));
                        const int cnt = lane < 8 ? ((word >> ((lane & 1) * 16)) & 0xffff) : 0;
                        int inc = cnt;
                        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);  // row_shr:1
                        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);  // row_shr:2
                        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);  // row_shr:4
                        const unsigned long long mj = __builtin_amdgcn_ballot_w64(lane < 8 && inc + ex > k);
                        const int j = (int)__builtin_ctzll(mj);
                        const int bin = ((wv_id << 6) + L) * 8 + j;
                        const int below = __builtin_amdgcn_readlane(inc - cnt, j) + ex;
                        const int cj = __builtin_amdgcn_readlane(cnt, j);
                        if (lane == 0) {
                            if (which == 0) {
                                sc->b1 = bin;
                                sc->below = below;
                                sc->c1 = cj;
                            } else {
                                sc->b2 = bin;
                                sc->c2 = cj;
                            }
                        }
                    }
                }
            }
        }
        ICV_SDP(10)
        // row offsets of cell k+1 (consumed in phase 3; requested here, ahead of the loads that phase 2 waits for, and
        // unconditionally: row 0 when there is no such cell)
        const bool has_next = nxt < P.n_rows;
        const int64_t nrow = has_next ? nxt : 0;
        const int64_t n0 = P.indptr[nrow + zoff], n1 = P.indptr[nrow + 1 + zoff];
        u32x4 g0a = {0u, 0u, 0u, 0u}, g0b = {0u, 0u, 0u, 0u};
        if (more) {
            g0a = __builtin_amdgcn_raw_buffer_load_b128(g0_rs, (unsigned)tl * 32u, 0, 0);
            g0b = __builtin_amdgcn_raw_buffer_load_b128(g0_rs, (unsigned)tl * 32u, 16, 0);
            reinterpret_cast<int4*>(hist)[tl] = make_int4(0, 0, 0, 0);  // every thread holds its 8 bins in hv
            int bad = 0;
            const auto add_entry = [&](const u32x4 e, float x) {
                if (e.x != 0xffffffffu) {  // masked columns contribute nothing
                    const float v = centre_clip<float>(x, __uint_as_float(e.y), __uint_as_float(e.z), cap, P.bounded, P.trunc);
                    const double d = (double)v - (double)__uint_as_float(e.w);
                    if (d != d) {
                        bad = 1;
                    } else {
                        // round to nearest integer, |d * scale| < 2^51: the low mantissa bits of d * scale + 1.5 * 2^52
                        const long long q = __double_as_longlong(d * scale + 6755399441055744.0) - 0x4338000000000000ll;
                        const int slot = (int)(e.x & 0xffffu);
                        atomicAdd(SQ + slot, (unsigned long long)q);
                        atomicAdd(SQ + slot + 1, (unsigned long long)(q * (long long)(e.x >> 16)));
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < kSdPF; ++i)
                if (tl + i * NT < c0) add_entry(etab[i], nval[i]);
            for (int k = tl + kSdPF * NT; k < c0; k += NT) {  // rows with more than 2048 stored entries
                const int g = P.indices[a0 + k];
                add_entry(reinterpret_cast<const u32x4*>(P.sd_tab)[g], static_cast<const float*>(P.values)[a0 + k]);
            }
            if (bad) sc->handback[par] = 1;  // benign race: every writer stores 1
        }
        ICV_SDP(2)
        __syncthreads();  // B1: bins complete; middle bins of the previous cell published
        ICV_SDP(3)
        asm volatile("" : "+v"(tl));

        // ---------------- phase 2: candidates gathered (previous cell); block totals (this cell) ----------------
        ICV_SD_PRIO(ICV_SD_P2)
        if (have_prev && sc->mode == 0) {
            const int b1 = sc->b1, b2 = sc->b2;
            const int n_in_bins = sc->c1 + (b2 != b1 ? sc->c2 : 0);
            if (n_in_bins <= 64) {
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    if (tl + i * NT < W) {
                        const int b = (int)((wbin[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
                        if (b == b1 || b == b2) {
                            const int idx = atomicAdd(&sc->ncand, 1);
                            if (idx < 64) sc->cand[idx] = wv[i];
                        }
                    }
                }
            } else if (tl == 0) {
                // too many windows share the median bins: hand the cell back to k_smooth
                const int slot = atomicAdd(P.row_count, 1);
                P.row_list[slot] = pcell;
                sc->ma = 0.0;
                sc->mb = 0.0;
                sc->mode = 2;
            }
        }
        ICV_SDP(11)
        if (more && wblk) {
            const int g0v[8] = {(int)g0a.x, (int)g0a.y, (int)g0a.z, (int)g0a.w, (int)g0b.x, (int)g0b.y, (int)g0b.z, (int)g0b.w};
            // {S0, T1 = sum of g d} of the thread's blocks 8 t .. 8 t + 7 (g: gene offset inside the chromosome); their
            // prefix sums over the blocks of this WAVEFRONT go back in place, the wavefront's total to the scratch:
            // a reader adds the totals of the wavefronts before (poff, formed in phase 3) -- nothing is carried in
            // registers across a barrier
            double s0[8], s1[8], t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int4 v = reinterpret_cast<const int4*>(SP)[k * kWsPlane + tl];
                const long long a = (long long)(((unsigned long long)(unsigned)v.y << 32) | (unsigned)v.x);
                const long long b = (long long)(((unsigned long long)(unsigned)v.w << 32) | (unsigned)v.z);
                s0[k] = (double)a;  // in units of 2^-k: the windows apply the scale (a power of two: exact)
                s1[k] = fma((double)g0v[k], s0[k], (double)b);
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
            if ((tl & 63) == 63) {
                sc->psum[tl >> 6] = y0;
                sc->psq[tl >> 6] = y1;
            }
        }
        ICV_SDP(4)
        __syncthreads();  // B2: candidates complete; wavefront totals published
        ICV_SDP(5)
        asm volatile("" : "+v"(tl));

        // ---------------- phase 3: median (previous cell); sums of the wavefronts before (this cell) -------------
        ICV_SD_PRIO(ICV_SD_P3)
        // Loads first: the tables of phase 4, then the entries of the next cell (needed two barriers later).  All of
        // them, and the x_res stores of phase 4, are unconditional buffer operations -- with an empty range where
        // there is nothing to do -- so that the compiler's counter bookkeeping stays exact and the waits in front of
        // the windows leave the prefetch and the stores in flight.
        int w_pack[MAXW], w_sr[MAXW];
        double wbase[MAXW];
        {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): everything older has been consumed or is consumed here
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {  // out-of-range windows read 0
                w_pack[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(wp_rs, (unsigned)tl * 4u, i * NT * 4, 0);
                w_sr[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(sr_rs, (unsigned)tl * 4u, i * NT * 4, 0);
                const u32x2 wb = __builtin_amdgcn_raw_buffer_load_b64(wb_rs, (unsigned)tl * 8u, i * NT * 8, 0);
                wbase[i] = __hiloint2double((int)wb.y, (int)wb.x);
            }
            // {column, value} of cell k+1
            const int64_t u0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(n0 >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)n0);
            const int64_t u1 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(n1 >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)n1);
            const int64_t cnt = has_next ? u1 - u0 : 0;
            const unsigned npf = (unsigned)(cnt < kSdPF * NT ? cnt : kSdPF * NT);
            const __amdgpu_buffer_rsrc_t i_rs = make_rsrc(P.indices + u0, npf * 4u);
            const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(static_cast<const float*>(P.values) + u0, npf * 4u);
#pragma unroll
            for (int i = 0; i < kSdPF; ++i) {
                nidx[i] = __builtin_amdgcn_raw_buffer_load_b32(i_rs, (unsigned)tl * 4u, i * NT * 4, 0);
                nval[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(v_rs, (unsigned)tl * 4u, i * NT * 4, 0));
            }
            a0 = u0;
            c0 = (int)(cnt < 0x7fffffff ? cnt : 0x7fffffff);
        }
        const int n_cand = sc->ncand;
        if (have_prev && sc->mode == 0 && (tl >> 6) * 8 < n_cand) {  // (a wavefront ranks candidates 8 w .. 8 w + 7)
            // exact float64 ranks of the <= 64 gathered candidates (see k_smooth_ws)
            const int n = n_cand < 64 ? n_cand : 64;
            const int below = sc->below;
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
        if (more && tl < 64) {
            // sums of the wavefronts before each one: lanes 0 .. 7 of wavefront 0, three DPP steps
            const bool in = tl < NWAVE;
            const double a = in ? sc->psum[tl & (NWAVE - 1)] : 0.0, b = in ? sc->psq[tl & (NWAVE - 1)] : 0.0;
            double ia = a, ib = b;
            ia += dpp_shift0<0x111, 0xf>(ia);
            ib += dpp_shift0<0x111, 0xf>(ib);
            ia += dpp_shift0<0x112, 0xf>(ia);
            ib += dpp_shift0<0x112, 0xf>(ib);
            ia += dpp_shift0<0x114, 0xf>(ia);
            ib += dpp_shift0<0x114, 0xf>(ib);
            if (in) sc->poff[tl] = make_double2(ia - a, ib - b);
        }
        ICV_SDP(6)
        __syncthreads();  // B3: wavefront offsets and the median of the previous cell published
        ICV_SDP(7)
        asm volatile("" : "+v"(tl));

        // ---------------- phase 4: x_res (previous cell); windows + histogram (this cell) -----------------------
        ICV_SD_PRIO(ICV_SD_P4)
        double sum = 0.0, sq = 0.0;
        const double med = (k1 == k2) ? sc->ma : (sc->ma + sc->mb) / 2.0;
        {
            // (a cell that was handed back gets med = 0 here and is rewritten by k_smooth afterwards)
            const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(P.out + pcell * P.ldo, have_prev ? (unsigned)W * 4u : 0u);
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                const double y = wv[i] - med;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)y), o_rs, (unsigned)tl * 4u, i * NT * 4, 0);
                if (tl + i * NT < W) {
                    sum = sum + y;
                    sq = fma(y, y, sq);
                }
            }
        }
        ICV_SDP(12)
        if (more) {
            // a pyramid window of n genes starting at gene s of its chromosome:
            //   [P1(m1) - P1(b-) - (s - 1)(P0(m1) - P0(b-))] + [(s + n)(P0(m2) - P0(m1)) - (P1(m2) - P1(m1))]
            // (b-: the block before the window, m1 / m2: last block of its first / second half), plus the same sum
            // of the zero row; a flat window is a difference of P0 (+ the zero row's sum) over the gene count
            const auto prefix_at = [&](int b) {  // blocks 0 .. b: prefix inside the wavefront + wavefronts before
                const double2 v = SP[ws_pidx(b)], o = sc->poff[b >> 9];
                return make_double2(v.x + o.x, v.y + o.y);
            };
            int lnan = 0;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                const int j = tl + i * NT;
                wv[i] = 0.0;
                if (j < W) {
                    const int wp = w_pack[i];
                    const int ln = wp >> 16, bs = wp & 0xffff;
                    const double2 pb = bs == 0 ? make_double2(0.0, 0.0) : prefix_at(bs - 1);
                    double v;
                    if (ln > 0) {
                        const double2 pm = prefix_at(bs + hbw - 1), pe = prefix_at(bs + nbw - 1);
                        const double sg = (double)w_sr[i];
                        const double a = (pm.y - pb.y) - (sg - 1.0) * (pm.x - pb.x);
                        const double d = (sg + (double)win) * (pe.x - pm.x) - (pe.y - pm.y);
                        v = finish_window(wbase[i] + (a + d) * qinv, ln, pyr_den, pyr_rcp, 1.0);
                    } else {  // flat: ln = -(padded genes of the chromosome)
                        const double pe = prefix_at(bs + (-ln) / B - 1).x;
                        v = (wbase[i] + (pe - pb.x) * qinv) / P.w_denom[j];
                    }
                    wv[i] = v;
                    lnan |= (v != v);
                    const int hb = hist_bin(v, inv_bound);
                    wbin[i >> 1] = (i & 1) ? (wbin[i >> 1] | ((unsigned)hb << 16)) : (unsigned)hb;
                    atomicAdd(&hist[hb >> 1], 1 << ((hb & 1) * 16));  // 16-bit bins, two per word
                }
            }
            if (lnan) sc->nanflag = 1;  // benign race: every writer stores 1
        }
        ICV_SDP(13)
        if (have_prev) {  // this wavefront's share of the moments of the previous cell, straight to HBM
            sum = wave_sum_dpp(sum);
            sq = wave_sum_dpp(sq);
            if ((tl & 63) == 0)
                reinterpret_cast<double2*>(P.cell_part)[pcell * NWAVE + (tl >> 6)] = make_double2(sum, sq);
            if (tl == 0) P.cell_median[pcell] = med;
        }
        ICV_SDP(8)
        __syncthreads();  // B4: histogram complete, prefix sums consumed
        ICV_SDP(9)
    }
#ifdef ICV_SD_PROFILE
    if (P.dbg && t == 64)
        for (int i = 0; i < 14; ++i) atomicAdd(P.dbg + i, sc->tacc[i]);
#endif
#undef ICV_SDP
}

}  // namespace icv

// k_smooth_ws: the fast smoothing kernel with a pipelined, iteration-free median
// (dense float32, blocked form).  Same L / S / W phases as k_smooth_fast, all 8 wavefronts take part;
// the median of cell k is resolved while cell k+1 flows through the same barriers:
//   W(k)          every window value also increments one bin of a 4096-bin LDS histogram
//                 (monotone piecewise-linear binning, so bins keep the rank order)
//   before A(k+1) every thread sums 8 bins, DPP prefix sums per wavefront, wavefront totals to LDS
//   after A(k+1)  the wavefront whose 512 bins hold a middle rank resolves its bin (b1, b2) and the number of
//                 windows in lower bins
//   after B1(k+1) every thread appends its windows of cell k that fall in b1 / b2 to cand[] (<= 64)
//   before B3(k+1) all wavefronts rank the candidates exactly in float64 -> median
//   after B3(k+1) x_res = window - median from the windows still in registers, store, moments
// Wave priority: the S, output and W phases (the serial chain of a cell) run at s_setprio 2, the scan and the L
// phase (table loads, scatter: throughput work) at 0 -- the co-resident workgroup fills the gaps (-8 % kernel time).
// A cell whose bins hold more than 64 candidates is handed back (row_list) and recomputed by the
// generic k_smooth right after this kernel; NaN cells are final here (median NaN).
// 5 workgroup barriers per cell (k_smooth_fast: ~10, with 2-3 data-dependent search rounds):
//   A  (histogram scanned, row free)        B1 (row scattered)
//   B2 (block sums in registers, candidates complete)
//   B3 ({S0,S1} in LDS, histogram cleared, median published)     B4 (histogram complete)
// Arithmetic and evaluation order of windows and median are those of k_smooth / k_smooth_fast
// (bit-identical x_res and moments).
#pragma once
#include "icv_kernels.hpp"

namespace icv {

constexpr int NBIN = 4096;  // 3072 bins over the central quarter of [-bound, bound], 512 per tail;
                            // stored as 16-bit counters, two per LDS word (8 KB)

struct ScratchW {
    int nanflag;
    int mode;               // 0: gather candidates of bins b1/b2, 1: median final, 2: cell handed back
    int b1, b2;
    int ncand;
    int below;              // windows in bins below b1
    int c1, c2;             // windows in bin b1 / b2
    int wtot[NWAVE];        // histogram scan: windows in the 512 bins scanned by each wavefront
    double ma, mb;          // the two middle order statistics of the previous cell, published before B3
    double psum[NWAVE], psq[NWAVE];  // per-wave moments of the previous cell
    double cand[64];
};
static_assert(sizeof(ScratchW) <= 1536, "ScratchW must fit the scratch region");

// LDS byte offset of the 16-bit position in the low / high half of a packed scatter-table word, in ONE VALU
// instruction (sub-dword operand select) instead of and/shift + shift; the row starts at LDS address 0
__device__ __forceinline__ unsigned lds_off_lo16(unsigned w, unsigned two) {
    unsigned a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(a) : "v"(two), "v"(w));
    return a;
}
__device__ __forceinline__ unsigned lds_off_hi16(unsigned w, unsigned two) {
    unsigned a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(a) : "v"(two), "v"(w));
    return a;
}
typedef __attribute__((address_space(3))) float lds_float_t;
#define ICV_LDS_F32_AT(OFF) (*reinterpret_cast<lds_float_t*>(static_cast<uintptr_t>(OFF)))

// Monotone non-decreasing map window value -> histogram bin.  Piecewise linear: |v| < bound/8 is
// resolved by 3072 bins (medians of centred, smoothed expression live there), the tails by 512 each.
__device__ __forceinline__ int hist_bin(double v, float inv_bound) {
    const float u = (float)v * inv_bound;  // in [-1, 1]
    // central segment first (almost every window of centred, smoothed expression): one fma, floor, two clamps
    if (__builtin_expect(fabsf(u) < 0.125f, 1)) {
        const int b = (int)floorf(fmaf(u, 3072.0f / 0.25f, 512.0f + 0.125f * (3072.0f / 0.25f)));
        return b < 512 ? 512 : (b > 3583 ? 3583 : b);  // segments stay disjoint under rounding
    }
    float f;
    int lo_b, hi_b;
    if (u < 0.0f) { f = (u + 1.0f) * (512.0f / 0.875f); lo_b = 0; hi_b = 511; }
    else { f = fmaf(u - 0.125f, 512.0f / 0.875f, 3584.0f); lo_b = 3584; hi_b = NBIN - 1; }
    const int b = (int)floorf(f);
    return b < lo_b ? lo_b : (b > hi_b ? hi_b : b);
}

// CSR input (k_csr_prepare has turned every stored entry into {LDS position, centred+clipped value}):
// the L phase writes the pre-centred zero row (clip(0 - ref), held in registers for the whole kernel)
// over the LDS row and scatters the cell's entries on top; the next cell's first PF entries per
// thread are prefetched like the dense row.
constexpr int kCsrPF = 4;  // prepared entries prefetched per thread (rows with <= 2048 entries; longer rows
                           // fetch the rest inside the L phase)

template <int UMAX, int MAXB, int MAXW, int BT, int NBW, bool CSR>
__global__ void __launch_bounds__(NT, 4) k_smooth_ws(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* row = reinterpret_cast<float*>(smem);
    double* S01 = reinterpret_cast<double*>(smem);
    int* hist = reinterpret_cast<int*>(smem + P.hist_off);
    ScratchW* sc = reinterpret_cast<ScratchW*>(smem + P.scratch_off);

    const int t = threadIdx.x;
    static_assert(NBIN == 8 * NT, "the histogram scan gives every thread 8 bins");
    const int W = P.W, NB = P.NB;
    const int B = BT > 0 ? BT : P.B;
    const int k1 = (W - 1) / 2, k2 = W / 2;
    const float inv_bound = (float)(1.0 / P.med_bound);
    const float cap = (float)P.cap;
    const unsigned row_bytes = (unsigned)P.n_cols * 4u;
    const unsigned voff = (unsigned)t * 16u, voff8 = (unsigned)t * 8u;
    const __amdgpu_buffer_rsrc_t lo_rs = make_rsrc(P.ref_lo, row_bytes);
    const __amdgpu_buffer_rsrc_t hi_rs = make_rsrc(P.bounded ? P.ref_hi : P.ref_lo, row_bytes);
    const __amdgpu_buffer_rsrc_t d16_rs = make_rsrc(P.dst16, (unsigned)(UMAX * NT * 8));
    // window descriptors (start block | length << 16): re-read from L2 in every L phase, ahead of the
    // row prefetch; as loop-carried registers they end up in scratch
    const __amdgpu_buffer_rsrc_t wp_rs = make_rsrc(P.w_pack, (unsigned)W * 4u);
    const double pyr_den = P.pyr_den, pyr_rcp = P.pyr_rcp;
    const float* xbase = static_cast<const float*>(P.values);
    // cells of this workgroup: blockIdx.x, +gridDim.x, ...; plus one pipeline-drain iteration
    const int64_t n_mine = (P.n_rows - blockIdx.x + gridDim.x - 1) / gridDim.x;

    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(smem)) != 0)
        __builtin_trap();  // the dense L phase addresses the row by absolute LDS offsets
    if (t == 0) {
        sc->nanflag = 0;
        sc->mode = 1;
        sc->ncand = 0;
        sc->ma = 0.0;
        sc->mb = 0.0;
    }
    u32x4 xq[CSR ? 1 : UMAX];  // dense: the prefetched row
    unsigned short epos[CSR ? kCsrPF : 1];  // CSR: prefetched entries of the next cell
    float eval[CSR ? kCsrPF : 1];
    // CSR: the pre-centred zero row is re-read from L2 in every L phase (it is not HBM traffic, and
    // holding it in 40 registers pushed the 8-blocks-per-thread variants into scratch)
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(P.zrow, (unsigned)P.zrow_bytes);
    if constexpr (!CSR) {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + (int64_t)blockIdx.x * P.ld, row_bytes);
#pragma unroll
        for (int u = 0; u < UMAX; ++u) xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * NT * 16, 0);
    } else {
        const int64_t e0 = P.indptr[blockIdx.x], e1 = P.indptr[blockIdx.x + 1];
#pragma unroll
        for (int i = 0; i < kCsrPF; ++i) {
            const int64_t k = e0 + t + i * NT;
            epos[i] = 0;
            eval[i] = 0.0f;
            if (k < e1) {
                epos[i] = P.pos16[k];
                eval[i] = P.cvals[k];
            }
        }
    }
// wave priority by phase (s_setprio): the arbiter of a SIMD prefers the wavefronts of the workgroup that is in
// its compute phases over the one that is issuing / waiting for loads
#ifndef ICV_PA
#define ICV_PA 0  // histogram scan
#endif
#ifndef ICV_PL
#define ICV_PL 0  // L phase
#endif
#ifndef ICV_PLI
#define ICV_PLI 0  // L phase while it issues loads
#endif
#ifndef ICV_PS
#define ICV_PS 2  // S phase, candidate gather
#endif
#ifndef ICV_PO
#define ICV_PO 2  // B2..B3 segment and x_res output
#endif
#ifndef ICV_PW
#define ICV_PW 2  // W phase
#endif
#ifndef ICV_W_INTERLEAVE
#define ICV_W_INTERLEAVE 1  // W phase: branch-free path for wavefronts whose windows are all full pyramid windows
#endif
#ifndef ICV_WGRP
#define ICV_WGRP 1  // windows of a thread advancing together in that path (2 spills a row vector: 2.2 ms)
#endif
#ifndef ICV_WCH
#define ICV_WCH 5  // {S0,S1} pairs of a window read per batch in the W phase (register budget)
#endif
#ifndef ICV_UH
#define ICV_UH 2  // reference / scatter-table vectors per load group of the L phase (divides UMAX)
#endif
    constexpr int UH = ICV_UH;
    static_assert(UMAX % UH == 0, "UMAX must be a multiple of UH");
    double wv[MAXW];  // this thread's windows of the previous cell (x_res needs its median)
    unsigned wbin[(MAXW + 1) / 2];  // their histogram bins, two 16-bit bins per register
#pragma unroll
    for (int i = 0; i < MAXW; ++i) wv[i] = 0.0;
#pragma unroll
    for (int i = 0; i < (MAXW + 1) / 2; ++i) wbin[i] = 0;
    __syncthreads();

    // phase timers cost ~18 VGPRs for every lane: compiled in only with -DICV_WS_PROFILE
#ifdef ICV_WS_PROFILE
    unsigned long long tlast = 0, tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ICV_PHASE(i)                                            \
    if (P.dbg && t == 64) {                                     \
        unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tacc[i] += now_ - tlast;                                \
        tlast = now_;                                           \
    }
    if (P.dbg && t == 64) tlast = __builtin_amdgcn_s_memtime();
#else
#define ICV_PHASE(i)
#endif

    for (int64_t it = 0; it <= n_mine; ++it) {
        const bool more = it < n_mine;  // a cell to smooth in this iteration
        const bool have_prev = it > 0;  // a cell whose median is being resolved
        const int64_t cell = (int64_t)blockIdx.x + it * gridDim.x;
        const int64_t pcell = cell - gridDim.x;
        int tl = t;
        asm volatile("" : "+v"(tl));  // see k_smooth_fast: keep thread-derived values out of LICM
        __builtin_amdgcn_s_setprio(ICV_PA);

        // ---------------- histogram scan of the previous cell, all wavefronts ----------------------
        // Level 1 before barrier A: thread g sums its 8 bins [8 g, 8 g + 8) (one ds_read_b128), a DPP prefix
        // sum inside the wavefront, the wavefront's total goes to LDS.  Level 2 after A, from registers (the
        // histogram itself is overwritten by the L phase): the wavefront whose 512 bins contain rank k1 (k2)
        // locates the lane and then the bin, and publishes it for the candidate gather after B1.
        int4 hv = make_int4(0, 0, 0, 0);
        int htot = 0, hincl = 0, nanf = 0;
        if (have_prev) {
            nanf = sc->nanflag;
            hv = reinterpret_cast<const int4*>(hist)[tl];
            const int s4 = (hv.x + hv.y) + (hv.z + hv.w);  // no carry between halves: counts <= W < 65536
            htot = (s4 & 0xffff) + ((unsigned)s4 >> 16);
            hincl = wave_scan_dpp(htot);
            if ((tl & 63) == 63) sc->wtot[tl >> 6] = hincl;
        }
        __syncthreads();  // A: histogram consumed, row free; wavefront totals published
        if (have_prev) {
            if (nanf) {
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
        ICV_PHASE(0)
        int w_pack[MAXW];
        if (ICV_PL != ICV_PA) __builtin_amdgcn_s_setprio(ICV_PL);
        if (more) {
          if constexpr (CSR) {
            // ---------------- L (CSR): zero row, then the cell's prepared entries ------------------
            const int nvec_row = P.scratch_off >> 4;  // 16-byte vectors of the LDS row region
            // row offsets of this cell and of the next one: requested first, so that their round trip overlaps the
            // zero-row copy
            const int64_t e0 = P.indptr[cell], e1 = P.indptr[cell + 1];
            const int64_t nxt = cell + gridDim.x;
            const int64_t n1 = nxt < P.n_rows ? P.indptr[nxt + 1] : e1;
            const int64_t n0 = nxt < P.n_rows ? P.indptr[nxt] : e1;
#pragma unroll
            for (int i = 0; i < MAXW; ++i)
                w_pack[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(wp_rs, (unsigned)tl * 4u, i * NT * 4, 0);
            // (the CSR variants have the registers for five zero-row vectors in flight: two L2 round trips per cell
            // instead of five)
            constexpr int UHC = (UMAX % 5 == 0) ? 5 : UH;
#pragma unroll
            for (int h = 0; h < UMAX; h += UHC) {
                u32x4 z[UHC];
#pragma unroll
                for (int k = 0; k < UHC; ++k) z[k] = __builtin_amdgcn_raw_buffer_load_b128(zr, voff, (h + k) * NT * 16, 0);
#pragma unroll
                for (int k = 0; k < UHC; ++k) {
                    const int i = (h + k) * NT + tl;
                    if (i < nvec_row) reinterpret_cast<u32x4*>(row)[i] = z[k];
                }
            }
            __syncthreads();  // every slot initialised before any entry lands on it
#pragma unroll
            for (int i = 0; i < kCsrPF; ++i)
                if (e0 + tl + i * NT < e1) row[epos[i]] = eval[i];
            for (int64_t k = e0 + tl + (int64_t)kCsrPF * NT; k < e1; k += NT) row[P.pos16[k]] = P.cvals[k];
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see the dense path
            if (nxt < P.n_rows) {
#pragma unroll
                for (int i = 0; i < kCsrPF; ++i) {
                    const int64_t k = n0 + tl + i * NT;
                    if (k < n1) {
                        epos[i] = P.pos16[k];
                        eval[i] = P.cvals[k];
                    }
                }
            }
          } else {
            // ---------------- L: centre, clip, scatter the prefetched row -------------------------
            // window descriptors first: they are the oldest loads of the phase, complete when the tables are
#pragma unroll
            for (int i = 0; i < MAXW; ++i)  // out-of-range windows read 0 (buffer bounds check)
                w_pack[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(wp_rs, (unsigned)tl * 4u, i * NT * 4, 0);
            for (int i = tl; i < P.n_pad; i += NT) row[P.pad_idx[i]] = 0.0f;
#define ICV_SCATTER4(X, LO, HI, D, BND)                                                                          \
    {                                                                                                            \
        const unsigned dx_ = (D).x, dy_ = (D).y;                                                                 \
        row[dx_ & 0xffffu] = centre_clip<float>(__uint_as_float((X).x), __uint_as_float((LO).x),                 \
                                                __uint_as_float((HI).x), cap, BND, P.trunc);                     \
        row[dx_ >> 16] = centre_clip<float>(__uint_as_float((X).y), __uint_as_float((LO).y),                     \
                                            __uint_as_float((HI).y), cap, BND, P.trunc);                         \
        row[dy_ & 0xffffu] = centre_clip<float>(__uint_as_float((X).z), __uint_as_float((LO).z),                 \
                                                __uint_as_float((HI).z), cap, BND, P.trunc);                     \
        row[dy_ >> 16] = centre_clip<float>(__uint_as_float((X).w), __uint_as_float((LO).w),                     \
                                            __uint_as_float((HI).w), cap, BND, P.trunc);                         \
    }
            if (!P.bounded) {
                // one reference row: y = x - ref, clip with v_med3_f32.  v_med3 drops NaNs, np.clip keeps
                // them: pairs are checked with an unordered compare and a fix-up pass (never taken on real
                // data) rewrites the NaN slots.
                bool any_nan = false;
                unsigned two = 2u;
                asm volatile("" : "+v"(two));  // a VGPR operand for the SDWA shifts
#pragma unroll
                for (int h = 0; h < UMAX; h += UH) {
                    u32x4 lo[UH];
                    u32x2 dd[UH];
                    if (ICV_PLI != ICV_PL) __builtin_amdgcn_s_setprio(ICV_PLI);
#pragma unroll
                    for (int k = 0; k < UH; ++k) {
                        lo[k] = __builtin_amdgcn_raw_buffer_load_b128(lo_rs, voff, (h + k) * NT * 16, 0);
                        dd[k] = __builtin_amdgcn_raw_buffer_load_b64(d16_rs, voff8, (h + k) * NT * 8, 0);
                    }
                    if (ICV_PLI != ICV_PL) __builtin_amdgcn_s_setprio(ICV_PL);
#pragma unroll
                    for (int k = 0; k < UH; ++k) {
                        const float y0 = __uint_as_float(xq[h + k].x) - __uint_as_float(lo[k].x);
                        const float y1 = __uint_as_float(xq[h + k].y) - __uint_as_float(lo[k].y);
                        const float y2 = __uint_as_float(xq[h + k].z) - __uint_as_float(lo[k].z);
                        const float y3 = __uint_as_float(xq[h + k].w) - __uint_as_float(lo[k].w);
                        any_nan |= __builtin_isunordered(y0, y1) | __builtin_isunordered(y2, y3);
                        ICV_LDS_F32_AT(lds_off_lo16(dd[k].x, two)) = __builtin_amdgcn_fmed3f(y0, -cap, cap);
                        ICV_LDS_F32_AT(lds_off_hi16(dd[k].x, two)) = __builtin_amdgcn_fmed3f(y1, -cap, cap);
                        ICV_LDS_F32_AT(lds_off_lo16(dd[k].y, two)) = __builtin_amdgcn_fmed3f(y2, -cap, cap);
                        ICV_LDS_F32_AT(lds_off_hi16(dd[k].y, two)) = __builtin_amdgcn_fmed3f(y3, -cap, cap);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (__builtin_expect(any_nan, 0)) {
#pragma unroll
                    for (int u = 0; u < UMAX; ++u) {
                        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(lo_rs, voff, u * NT * 16, 0);
                        const u32x2 dq = __builtin_amdgcn_raw_buffer_load_b64(d16_rs, voff8, u * NT * 8, 0);
                        const float y0 = __uint_as_float(xq[u].x) - __uint_as_float(lo.x);
                        const float y1 = __uint_as_float(xq[u].y) - __uint_as_float(lo.y);
                        const float y2 = __uint_as_float(xq[u].z) - __uint_as_float(lo.z);
                        const float y3 = __uint_as_float(xq[u].w) - __uint_as_float(lo.w);
                        if (y0 != y0) row[dq.x & 0xffffu] = y0;
                        if (y1 != y1) row[dq.x >> 16] = y1;
                        if (y2 != y2) row[dq.y & 0xffffu] = y2;
                        if (y3 != y3) row[dq.y >> 16] = y3;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UMAX; ++u) {
                    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(lo_rs, voff, u * NT * 16, 0);
                    const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(hi_rs, voff, u * NT * 16, 0);
                    const u32x2 dd = __builtin_amdgcn_raw_buffer_load_b64(d16_rs, voff8, u * NT * 8, 0);
                    ICV_SCATTER4(xq[u], lo, hi, dd, 1)
                }
            }
#undef ICV_SCATTER4
            // Every load issued so far has been consumed or is about to be (window descriptors): drain the
            // counter HERE, so that from now on only the row prefetch below is outstanding.  Without this the
            // compiler's conservative per-path bookkeeping puts s_waitcnt vmcnt(small) in front of later uses
            // of these registers (S, O and W phases), which can stall on the HBM prefetch and on the stores.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
            const int64_t nxt = cell + gridDim.x;
            if (ICV_PLI != ICV_PL) __builtin_amdgcn_s_setprio(ICV_PLI);
            if (nxt < P.n_rows) {
                const __amdgpu_buffer_rsrc_t xr = make_rsrc(xbase + nxt * P.ld, row_bytes);
#pragma unroll
                for (int u = 0; u < UMAX; ++u) xq[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, u * NT * 16, 0);
            }
          }
        }
        ICV_PHASE(1)
        __syncthreads();  // B1: row scattered
        ICV_PHASE(2)
        asm volatile("" : "+v"(tl));
        if (have_prev && !nanf) {
            const int b1 = sc->b1, b2 = sc->b2;
            const int n_in_bins = sc->c1 + (b2 != b1 ? sc->c2 : 0);
            if (n_in_bins <= 64) {
                // windows of the previous cell in the bins of its two middle order statistics -> cand[]
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
        // ---------------- S: block partial sums (registers) ---------------------------------------
        double s0[MAXB], s1[MAXB];
        __builtin_amdgcn_s_setprio(ICV_PS);
        if (more) {
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                s0[i] = 0.0;
                s1[i] = 0.0;
                const int b = tl + i * NT;
                if (b < NB) {
                    const float* rp = row + b * B;
                    if constexpr (BT > 0 && (BT & 1) == 0) {
                        const float2* rp2 = reinterpret_cast<const float2*>(rp);
#pragma unroll
                        for (int r = 0; r < BT; r += 2) {
                            const float2 v2 = rp2[r >> 1];
                            block_accumulate((double)v2.x, r, s0[i], s1[i]);
                            block_accumulate((double)v2.y, r + 1, s0[i], s1[i]);
                        }
                    } else if constexpr (BT > 0) {
#pragma unroll
                        for (int r = 0; r < BT; ++r) block_accumulate((double)rp[r], r, s0[i], s1[i]);
                    } else {
#pragma unroll 1
                        for (int r = 0; r < B; ++r) block_accumulate((double)rp[r], r, s0[i], s1[i]);
                    }
                }
            }
        }
        ICV_PHASE(3)
        __syncthreads();  // B2: row dead; candidates of the previous cell complete
        asm volatile("" : "+v"(tl));
        if (ICV_PO != ICV_PS) __builtin_amdgcn_s_setprio(ICV_PO);
        if (more) {
            int4* h4 = reinterpret_cast<int4*>(hist);  // clear the histogram (dead part of the row)
            for (int i = tl; i < NBIN / 8; i += NT) h4[i] = make_int4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int b = tl + i * NT;
                if (b < NB) *reinterpret_cast<double2*>(S01 + 2 * b) = make_double2(s0[i], s1[i]);
            }
        }
        if (have_prev && sc->mode == 0) {
            // exact float64 ranks of the <= 64 gathered candidates, all wavefronts: thread (ci, part)
            // compares candidate ci with candidates 8 part .. 8 part + 7, the 8 partial ranks of a
            // candidate sit in 8 consecutive lanes and are added with DPP
            const int n = sc->ncand < 64 ? sc->ncand : 64;
            const int below = sc->below;
            const int ci = tl >> 3, part = tl & 7;
            // branch-free: all reads unconditional (cand[] is always allocated), 4 x ds_read_b128 in flight
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
        __syncthreads();  // B3: {S0,S1} ready, histogram cleared, median of the previous cell published
        ICV_PHASE(4)
        asm volatile("" : "+v"(tl));
        if (have_prev) {
            // x_res of the previous cell from the windows still in registers (a cell that was handed back gets
            // med = 0 here and is rewritten, with its median and moments, by k_smooth afterwards)
            const double med = (k1 == k2) ? sc->ma : (sc->ma + sc->mb) / 2.0;
            double sum = 0.0, sq = 0.0;
            float* orow = P.out + pcell * P.ldo;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                const int j = tl + i * NT;
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
                reinterpret_cast<double2*>(P.cell_part)[pcell * NWAVE + (tl >> 6)] = make_double2(sum, sq);
            if (tl == 0) P.cell_median[pcell] = med;
        }
        // ---------------- W: windows (registers) + histogram --------------------------------------
        if (ICV_PW != ICV_PO) __builtin_amdgcn_s_setprio(ICV_PW);
        bool w_done = false;
#if ICV_W_INTERLEAVE
        if constexpr (BT > 0 && NBW > 0 && NBW % 2 == 0 && NBW <= 10) {
            // Wavefronts whose windows are all full pyramid windows (every one in the benchmark geometry) take a
            // path without per-window branches: a thread's missing last window repeats its first one and is
            // discarded; the order inside a window is the canonical one.  ICV_WGRP windows advance together
            // (interleaved float64 chains); more than one does not fit the registers next to the prefetched row.
            bool ok = more;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {  // in place: the per-window path below never looks at a missing window
                if (i > 0) w_pack[i] = (tl + i * NT < W) ? w_pack[i] : w_pack[0];
                ok &= (w_pack[i] >> 16) == NBW * BT;
            }
            if (more && __builtin_amdgcn_ballot_w64(!ok) == 0) {
                constexpr int HB = NBW / 2, GW = ICV_WGRP;  // windows advancing together (register budget)
                static_assert(MAXW % GW == 0, "MAXW must be a multiple of the window group");
                int lnan = 0;
#pragma unroll
                for (int g0 = 0; g0 < MAXW; g0 += GW) {
                    const double2* sp[GW];
                    double v[GW];
#pragma unroll
                    for (int i = 0; i < GW; ++i) {
                        sp[i] = reinterpret_cast<const double2*>(S01) + (w_pack[g0 + i] & 0xffff);
                        v[i] = 0.0;
                    }
#pragma unroll
                    for (int m = 0; m < NBW; ++m) {
                        double2 sv[GW];
#pragma unroll
                        for (int i = 0; i < GW; ++i) sv[i] = sp[i][m];
#pragma unroll
                        for (int i = 0; i < GW; ++i)
                            v[i] = fma((double)(m < HB ? m * BT + 1 : NBW * BT - m * BT), sv[i].x, v[i]);
#pragma unroll
                        for (int i = 0; i < GW; ++i) v[i] = m < HB ? v[i] + sv[i].y : v[i] - sv[i].y;
                        if (m % ICV_WCH == ICV_WCH - 1) __builtin_amdgcn_sched_barrier(0);  // bound the reads in flight
                    }
#pragma unroll
                    for (int i = 0; i < GW; ++i) v[i] = finish_window(v[i], NBW * BT, pyr_den, pyr_rcp, 1.0);
#pragma unroll
                    for (int i = 0; i < GW; ++i) {
                        const int ii = g0 + i;
                        const bool valid = tl + ii * NT < W;
                        wv[ii] = valid ? v[i] : 0.0;
                        lnan |= valid & (v[i] != v[i]);
                        __builtin_amdgcn_sched_barrier(0);
                        const int hb = hist_bin(v[i], inv_bound);
                        wbin[ii >> 1] = (ii & 1) ? (wbin[ii >> 1] | ((unsigned)hb << 16)) : (unsigned)hb;
                        if (valid) atomicAdd(&hist[hb >> 1], 1 << ((hb & 1) * 16));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (lnan) sc->nanflag = 1;
                w_done = true;
            }
        }
#endif
        if (more && !w_done) {
            int lnan = 0;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                const int j = tl + i * NT;
                wv[i] = 0.0;
                if (j < W) {
                    const int wp = w_pack[i];
                    const int ln = wp >> 16;
                    const double2* sp = reinterpret_cast<const double2*>(S01) + (wp & 0xffff);
                    double v;
                    if constexpr (BT > 0 && NBW > 0) {
                        if (ln == NBW * BT) {
                            // same operation order as window_from_blocks, fully unrolled in batches of five
                            // {S0,S1} pairs (only 5 x ds_read_b128 results live at a time: register budget)
                            constexpr int HB = NBW / 2, CH = ICV_WCH;
                            v = 0.0;
#pragma unroll
                            for (int m0 = 0; m0 < NBW; m0 += CH) {
                                double2 sb[CH];
#pragma unroll
                                for (int u = 0; u < CH; ++u)
                                    if (m0 + u < NBW) sb[u] = sp[m0 + u];
#pragma unroll
                                for (int u = 0; u < CH; ++u) {
                                    const int m = m0 + u;
                                    if (m < NBW) {
                                        if (m < HB) {
                                            v = fma((double)(m * BT + 1), sb[u].x, v);
                                            v = v + sb[u].y;
                                        } else {
                                            v = fma((double)(NBW * BT - m * BT), sb[u].x, v);
                                            v = v - sb[u].y;
                                        }
                                    }
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
                            v = window_from_blocks(ln, B, [&](int m, double& a, double& b2) {
                                const double2 s = sp[m];
                                a = s.x;
                                b2 = s.y;
                            });
                        }
                    } else {
                        v = window_from_blocks(ln, B, [&](int m, double& a, double& b2) {
                            const double2 s = sp[m];
                            a = s.x;
                            b2 = s.y;
                        });
                    }
                    // flat windows (one per chromosome with <= window genes) read their gene count from
                    // the window table: a rare global load in this otherwise load-free phase
                    v = finish_window(v, ln, pyr_den, pyr_rcp, ln > 0 ? 1.0 : P.w_denom[j]);
                    wv[i] = v;
                    lnan |= (v != v);
                    const int hb = hist_bin(v, inv_bound);
                    wbin[i >> 1] = (i & 1) ? (wbin[i >> 1] | ((unsigned)hb << 16)) : (unsigned)hb;
                    atomicAdd(&hist[hb >> 1], 1 << ((hb & 1) * 16));  // 16-bit bins, two per word
                }
            }
            if (lnan) sc->nanflag = 1;  // benign race: every writer stores 1
        }
        __syncthreads();  // B4: histogram complete; moments of the previous cell complete
        ICV_PHASE(5)
    }
#ifdef ICV_WS_PROFILE
    if (P.dbg && t == 64)
        for (int i = 0; i < 8; ++i) atomicAdd(P.dbg + i, tacc[i]);
#endif
#undef ICV_PHASE
}

}  // namespace icv

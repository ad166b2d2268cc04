// calculate_gene_values (reference tl/_infercnv.py:247-298, :443-453) from the float64 windows the smoothing kernel
// already produced -- ONE kernel, one pass over the windows, one write of the cells x genes float64 layer.
//
//   gene value   = np.mean over the kept windows that contain the gene (:274-288; numpy's pairwise float64 sum)
//   centring     = minus the per-cell np.median over the covered genes (:443-444)
//   noise filter = |v| < the chunk's threshold -> 0 (:452-453, the same thr as X_cnv)
//   genes without a value: NaN (the reference's reindex with fill_value=np.nan, :147)
//
// A gene's value depends only on WHICH windows cover it, (j0, cnt): consecutive covered genes share it in runs of
// ~step genes (icv_plan.hpp: gv_run_*: ~2 G / step runs).  Per cell: the W windows come into LDS once, every run's
// value is formed once (numpy's summation order), the median over the covered genes is the WEIGHTED median of the run
// values (weights = genes per run: the same multiset the reference sorts), and the output row is written once, in
// input-column order, as full 16-byte stores (NaN where a column has no run).  The round-1 form (k_gene_means ->
// k_row_median -> k_gene_finish on a cells x n_cov float64 temporary after a NaN fill of the whole layer) wrote the
// layer twice and read the temporary ~20 times: 84 ms per 100 000 x 20 000 cells against the 3.3 ms of the plain call.
//
// Weighted median = an exact order statistic: a 2048-bin weighted histogram over [min, max] of the cell's run values
// locates the bin of rank k, the bin's elements (<= 64; else the histogram is refined on the bin's own [min, max]) are
// ranked exactly by one wavefront.  (First version: 8-pass radix select on the 64-bit keys -- its top byte is shared by
// all values of one sign: 32-way same-address LDS atomics, 30 000 cycles per cell.)
#pragma once
#include "icv_kernels.hpp"

namespace icv {

constexpr int kGvThreads = 512;
constexpr int kGvWaves = kGvThreads / 64;
constexpr int kGvBins = 4 * kGvThreads;  // every thread scans four bins (256 / 384 threads x 3 workgroups per CU measured slower: 9.7 / 8.5 against 7.3 ms per call)
constexpr int kGvCand = 64;

struct GvScratch {
    unsigned hist[kGvBins];
    double red[2][kGvWaves];   // min / max partials
    double cand_v[kGvCand];
    int cand_w[kGvCand];
    int wsum[kGvWaves];        // per-wavefront histogram totals (scan)
    int ncand, anynan, sel_bin, sel_below;
    double v1, v2;
    int found2, pad_;
};

// (the tables need W <= 65 535 windows, run lengths and window counts that fit 16 bits, fewer than 65 535 runs)
// dynamic LDS: win[W] (float64; dead after the run values: the scratch aliases it) | val[R + 2] (val[R] = NaN: the slot of
// the columns without a value) | mult[R] (8 / 16-bit) | the run table (32-bit) where it fits
// (run lengths as bytes where no run is longer than 255 genes: 4 KB less at 4 000 runs -- three workgroups per CU instead of two)
// pk_lds: the run table {first window | count << 16} in LDS as well (4 R bytes): no load from memory between a cell's
// stores and the next cell's run values (see the kernel)
__host__ __device__ inline size_t gv_lds_bytes(int W, int R, int mult_bytes = 2, bool pk_lds = false) {
    size_t w = (size_t)W * 8;
    if (w < sizeof(GvScratch)) w = sizeof(GvScratch);
    w = (w + 15) / 16 * 16;
    return w + ((size_t)R + 2) * 8 + ((size_t)R * mult_bytes + 15) / 16 * 16 + (pk_lds ? (size_t)R * 4 : 0);
}

// numpy's float64 add.reduce of a[0..n) in LDS for n <= 128 (the launcher sends plans with longer runs -- a gene covered
// by more than 128 kept windows: step 1 with a window beyond 128 -- to the round-1 kernels, whose numpy_sum recurses: the
// call alone cost this kernel a 192-byte stack frame)
__device__ __forceinline__ double gv_numpy_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int i = 8;
        for (; i + 8 <= n; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    return __builtin_nan("");  // (never reached: gv_fused_ok)
}

// run_pk[r] = first window | #windows << 16 of run r; col_run16[c] = run of input column c, or R -- the NaN slot val[R] --
// where the gene has no value (padded to a multiple of 8 columns with R).  Global loads are issued in batches before the
// dependent LDS work.
__global__ void __launch_bounds__(kGvThreads) k_gene_fused(
    const double* __restrict__ win, int64_t ldw, int64_t n_rows, int W, const uint32_t* __restrict__ run_pk,
    const int32_t* __restrict__ run_mult, int R, int n_cov, const uint16_t* __restrict__ col_run16, int n_cols,
    const double* __restrict__ thr, int64_t chunksize, int64_t row_phase, double* __restrict__ out, int64_t ldg,
    int mult_bytes, int pk_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    size_t woff = (size_t)W * 8;
    if (woff < sizeof(GvScratch)) woff = sizeof(GvScratch);
    woff = (woff + 15) / 16 * 16;
    double* lwin = reinterpret_cast<double*>(gsm);
    GvScratch* sc = reinterpret_cast<GvScratch*>(gsm);  // aliases the windows (used after they are dead)
    double* val = reinterpret_cast<double*>(gsm + woff);
    const size_t vbytes = ((size_t)R + 2) * 8;
    unsigned short* mult16 = reinterpret_cast<unsigned short*>(gsm + woff + vbytes);
    unsigned char* mult8 = reinterpret_cast<unsigned char*>(mult16);
    uint32_t* lpk = reinterpret_cast<uint32_t*>(gsm + woff + vbytes + ((size_t)R * mult_bytes + 15) / 16 * 16);
    const bool m8 = mult_bytes == 1;
    const auto mult_of = [&](int r) { return m8 ? (int)mult8[r] : (int)mult16[r]; };
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int k1 = (n_cov - 1) / 2, k2 = n_cov / 2;
    for (int i = t; i < R; i += kGvThreads) {
        if (m8) mult8[i] = (unsigned char)run_mult[i];
        else mult16[i] = (unsigned short)run_mult[i];
        if (pk_lds) lpk[i] = run_pk[i];
    }
    if (t == 0) val[R] = __builtin_nan("");  // what a column without a run reads (the table says R there)
    const bool vec2 = (ldg % 2 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const bool win16 = (ldw % 2 == 0) && ((reinterpret_cast<uintptr_t>(win) & 15) == 0);
    const double nan = __builtin_nan("");
    constexpr int UW = 4;  // window loads (16 bytes) in flight per thread
    constexpr int UR = 4;  // run-table loads in flight per thread
    constexpr int UC = 20;  // column-table loads (2 columns each) in flight per thread: a 20 480-column row in ONE batch
    constexpr int PW = 2;  // 16-byte window loads per thread prefetched a cell ahead (W <= 2 048: all of them)
    // gfx9 counts a wavefront's loads AND stores in one in-order counter: a load issued after the row's stores is only
    // there when the stores are acknowledged -- the cell's 160 KB drain before the next cell can even start.  So nothing
    // is loaded after a store: the NEXT cell's windows and all of the thread's column-table words are fetched before the
    // first store of the output phase, the stores go out back to back, and they drain while the next cell's run values
    // and median are formed from LDS.  (The table words live in registers for the output phase only; kept across cells
    // the compiler took 252 VGPRs.)
    typedef double f64x2_t __attribute__((ext_vector_type(2)));
    f64x2_t pre[PW];
    const int n2w = W / 2;
    const auto prefetch = [&](int64_t cell_) {
        const double* wr_ = win + cell_ * ldw;
#pragma unroll
        for (int u = 0; u < PW; ++u)
            if (t + u * kGvThreads < n2w) pre[u] = *reinterpret_cast<const f64x2_t*>(wr_ + 2 * (t + u * kGvThreads));
    };
    if (win16 && (int64_t)blockIdx.x < n_rows) prefetch(blockIdx.x);
    for (int64_t cell = blockIdx.x; cell < n_rows; cell += gridDim.x) {
        __syncthreads();  // the previous cell's output phase has read val[]; its scratch use is over
        // ---- 1. the cell's windows
        const double* wr = win + cell * ldw;
        if (win16) {
#pragma unroll
            for (int u = 0; u < PW; ++u)
                if (t + u * kGvThreads < n2w) *reinterpret_cast<f64x2_t*>(lwin + 2 * (t + u * kGvThreads)) = pre[u];
            const int n2 = W / 2;
            for (int j0 = t + PW * kGvThreads; j0 < n2; j0 += UW * kGvThreads) {
                f64x2_t q[UW];
#pragma unroll
                for (int u = 0; u < UW; ++u)
                    if (j0 + u * kGvThreads < n2) q[u] = *reinterpret_cast<const f64x2_t*>(wr + 2 * (j0 + u * kGvThreads));
#pragma unroll
                for (int u = 0; u < UW; ++u)
                    if (j0 + u * kGvThreads < n2) *reinterpret_cast<f64x2_t*>(lwin + 2 * (j0 + u * kGvThreads)) = q[u];
            }
            if ((W & 1) && t == 0) lwin[W - 1] = wr[W - 1];
        } else {
            for (int j = t; j < W; j += kGvThreads) lwin[j] = wr[j];
        }
        __syncthreads();
        // ---- 2. run values (np.mean of the covering windows: numpy's pairwise sum / count); min / max / NaN
        int nanl = 0;
        double mn = __builtin_inf(), mx = -__builtin_inf();
        for (int r0 = t; r0 < R; r0 += UR * kGvThreads) {
            uint32_t pk[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u)
                pk[u] = r0 + u * kGvThreads < R ? (pk_lds ? lpk[r0 + u * kGvThreads] : run_pk[r0 + u * kGvThreads]) : 0u;
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int r = r0 + u * kGvThreads;
                if (r < R) {
                    const int cnt = (int)(pk[u] >> 16);
                    const double v = gv_numpy_sum(lwin + (pk[u] & 0xffffu), cnt) / (double)cnt;
                    val[r] = v;
                    nanl |= (v != v);
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
            }
        }
        mn = wave_min_dpp(mn);
        mx = wave_max_dpp(mx);
        __syncthreads();  // windows dead: the scratch may be written
        if (t == 0) {
            sc->anynan = 0;
            sc->found2 = 0;
        }
        if (lane == 0) {
            sc->red[0][wave] = mn;
            sc->red[1][wave] = mx;
        }
        __syncthreads();
        if (nanl) sc->anynan = 1;  // benign race
        double lo = sc->red[0][0], hi = sc->red[1][0];
#pragma unroll
        for (int i = 1; i < kGvWaves; ++i) {
            lo = sc->red[0][i] < lo ? sc->red[0][i] : lo;
            hi = sc->red[1][i] > hi ? sc->red[1][i] : hi;
        }
        __syncthreads();
        const bool anynan = sc->anynan != 0;
        // ---- 3. weighted median: rank k1 (and k2 = k1 + 1 for an even count) of the multiset {val[r] x mult[r]}
        double v1 = lo, v2 = lo;
#if defined(ICV_DEV_EXPERIMENTS) && defined(GV_EXP_NOMEDIAN)
        if (false) {
#else
        if (!anynan && R > 0) {
#endif
            int below = 0;  // weight strictly below the current [lo, hi]
            bool need2 = k2 != k1;
            for (int level = 0; level < 64; ++level) {
                if (!(hi > lo)) {  // every remaining element equals lo: ranks k1 (and k2 if it is inside) are lo
                    v1 = lo;
                    // weight of the elements equal to lo
                    int w = 0;
                    for (int r = t; r < R; r += kGvThreads) w += (val[r] == lo) ? mult_of(r) : 0;
                    w = wave_sum_i(w);
                    if (lane == 0) sc->wsum[wave] = w;
                    __syncthreads();
                    int wt = 0;
#pragma unroll
                    for (int i = 0; i < kGvWaves; ++i) wt += sc->wsum[i];
                    if (need2 && k2 - below < wt) {
                        v2 = lo;
                        need2 = false;
                    }
                    break;
                }
                // histogram of the elements in [lo, hi]
                for (int i = t; i < kGvBins; i += kGvThreads) sc->hist[i] = 0u;
                if (t == 0) sc->ncand = 0;
                __syncthreads();
                const double scale = (double)kGvBins / (hi - lo);
                for (int r = t; r < R; r += kGvThreads) {
                    const double v = val[r];
                    if (v >= lo && v <= hi) {
                        int b = (int)((v - lo) * scale);
                        b = b > kGvBins - 1 ? kGvBins - 1 : b;
                        atomicAdd(&sc->hist[b], (unsigned)mult_of(r));
                    }
                }
                __syncthreads();
                // the bin of rank k1 - below: every thread sums its four bins, wavefront scan, wavefront totals
                const int kk = k1 - below;
                const uint4 h = reinterpret_cast<const uint4*>(sc->hist)[t];
                const int c = (int)((h.x + h.y) + (h.z + h.w));
                const int incl = wave_scan_dpp(c);
                if (lane == 63) sc->wsum[wave] = incl;
                __syncthreads();
                int wbase = 0;
#pragma unroll
                for (int i = 0; i < kGvWaves; ++i) wbase += i < wave ? sc->wsum[i] : 0;
                const int excl = wbase + incl - c;
                if (kk >= excl && kk < excl + c) {  // exactly one thread
                    int run = excl, d = 3;
                    if (kk < run + (int)h.x) d = 0;
                    else if (kk < run + (int)(h.x + h.y)) { d = 1; run += (int)h.x; }
                    else if (kk < run + (int)(h.x + h.y + h.z)) { d = 2; run += (int)(h.x + h.y); }
                    else run += (int)(h.x + h.y + h.z);
                    sc->sel_bin = 4 * t + d;
                    sc->sel_below = run;
                }
                __syncthreads();
                const int sb = sc->sel_bin;
                below += sc->sel_below;
                // the elements of that bin: candidates (<= 64) or the next level's range
                double bmn = __builtin_inf(), bmx = -__builtin_inf();
                for (int r = t; r < R; r += kGvThreads) {
                    const double v = val[r];
                    if (v >= lo && v <= hi) {
                        int b = (int)((v - lo) * scale);
                        b = b > kGvBins - 1 ? kGvBins - 1 : b;
                        if (b == sb) {
                            const int idx = atomicAdd(&sc->ncand, 1);
                            if (idx < kGvCand) {
                                sc->cand_v[idx] = v;
                                sc->cand_w[idx] = mult_of(r);
                            }
                            bmn = v < bmn ? v : bmn;
                            bmx = v > bmx ? v : bmx;
                        }
                    }
                }
                bmn = wave_min_dpp(bmn);
                bmx = wave_max_dpp(bmx);
                if (lane == 0) {
                    sc->red[0][wave] = bmn;
                    sc->red[1][wave] = bmx;
                }
                __syncthreads();
                const int nc = sc->ncand;
                if (nc <= kGvCand) {
                    if (wave == 0) {
                        // exact ranks inside the bin: weight below / equal for every candidate
                        const double mine = lane < nc ? sc->cand_v[lane] : __builtin_inf();
                        int wl = 0, we = 0;
                        for (int q = 0; q < nc; ++q) {
                            const double o = sc->cand_v[q];
                            const int w = sc->cand_w[q];
                            wl += o < mine ? w : 0;
                            we += o == mine ? w : 0;
                        }
                        const int ka = k1 - below, kb = k2 - below;
                        if (lane < nc && ka >= wl && ka < wl + we) sc->v1 = mine;  // (equal candidates write the same value)
                        if (lane < nc && kb >= wl && kb < wl + we) {
                            sc->v2 = mine;
                            sc->found2 = 1;
                        }
                    }
                    __syncthreads();
                    v1 = sc->v1;
                    if (need2 && sc->found2) {
                        v2 = sc->v2;
                        need2 = false;
                    }
                    break;
                }
                lo = sc->red[0][0];
                hi = sc->red[1][0];
#pragma unroll
                for (int i = 1; i < kGvWaves; ++i) {
                    lo = sc->red[0][i] < lo ? sc->red[0][i] : lo;
                    hi = sc->red[1][i] > hi ? sc->red[1][i] : hi;
                }
                __syncthreads();
            }
            if (need2) {
                // rank k2 is the smallest value above v1
                double m2 = __builtin_inf();
                for (int r = t; r < R; r += kGvThreads) {
                    const double v = val[r];
                    if (v > v1 && v < m2) m2 = v;
                }
                m2 = wave_min_dpp(m2);
                __syncthreads();
                if (lane == 0) sc->red[0][wave] = m2;
                __syncthreads();
                v2 = sc->red[0][0];
#pragma unroll
                for (int i = 1; i < kGvWaves; ++i) v2 = sc->red[0][i] < v2 ? sc->red[0][i] : v2;
            } else if (k2 == k1) {
                v2 = v1;
            }
        }
        const double med = anynan ? nan : ((k1 == k2) ? v1 : (v1 + v2) / 2.0);
        // ---- 4. the output row, input-column order: value - median, noise filter, NaN where there is no value.
        // The arithmetic is done once per RUN, in place (4 000 runs, not 20 000 columns: the per-column form spent 40 % of
        // the kernel's VALU instructions here); a column then is one LDS read at its table index (val[R] = NaN).
        const bool has_thr = thr != nullptr;
        const double th = has_thr ? thr[(cell + row_phase) / chunksize] : 0.0;
        double* orow = out + cell * ldg;
        for (int r = t; r < R; r += kGvThreads) {
            double a = val[r] - med;
            if (has_thr && fabs(a) < th) a = 0.0;
            val[r] = a;
        }
        __syncthreads();
        const auto value_of = [&](unsigned rx) { return val[rx]; };
        if (vec2) {
            // a lane = two adjacent columns, the lanes of a wavefront = 1 KB of the row: every store instruction writes
            // whole lines (eight columns per lane -- 64-byte runs, lane stride 64 B -- measured 2.4 x slower: partial lines)
            const int n2 = (n_cols + 1) / 2;  // pairs (the table is padded with R)
            const uint32_t* cr2 = reinterpret_cast<const uint32_t*>(col_run16);
            for (int g0 = t; g0 < n2; g0 += UC * kGvThreads) {
                uint32_t rr[UC];
#pragma unroll
                for (int u = 0; u < UC; ++u) rr[u] = g0 + u * kGvThreads < n2 ? cr2[g0 + u * kGvThreads] : 0u;
                // (the last loads before this cell's stores: the next cell's windows)
                if (g0 + UC * kGvThreads >= n2 && win16 && cell + gridDim.x < n_rows) prefetch(cell + gridDim.x);
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int c = 2 * (g0 + u * kGvThreads);
                    if (c >= n_cols) continue;
                    const double a = value_of(rr[u] & 0xffffu), b = value_of(rr[u] >> 16);
                    if (c + 2 <= n_cols) {
                        const f64x2_t q = {a, b};
#if defined(ICV_DEV_EXPERIMENTS) && defined(GV_EXP_PLAIN)
                        *reinterpret_cast<f64x2_t*>(orow + c) = q;
#elif defined(ICV_DEV_EXPERIMENTS) && defined(GV_EXP_NOSTORE)
                        if (a == 1.2345e300) *reinterpret_cast<f64x2_t*>(orow + c) = q;
#else
                        __builtin_nontemporal_store(q, reinterpret_cast<f64x2_t*>(orow + c));
#endif
                    } else {
                        orow[c] = a;
                    }
                }
            }
        } else {
            if (win16 && cell + gridDim.x < n_rows) prefetch(cell + gridDim.x);
            for (int c = t; c < n_cols; c += kGvThreads) orow[c] = value_of((unsigned)col_run16[c]);
        }
    }
}

}  // namespace icv

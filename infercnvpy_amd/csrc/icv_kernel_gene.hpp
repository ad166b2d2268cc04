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
// values (weights = genes per run: the same multiset the reference sorts) found by an 8-pass radix select on the
// order-preserving 64-bit keys, and the output row is written once, in input-column order, as full 16-byte stores
// (NaN where a column has no run).  The round-1 form (k_gene_means -> k_row_median -> k_gene_finish on a cells x n_cov
// float64 temporary after a NaN fill of the whole layer) wrote the layer twice and read the temporary ~20 times.
#pragma once
#include "icv_kernels.hpp"

namespace icv {

constexpr int kGvThreads = 256;

struct GvScratch {
    int hist[256];
    int sel_digit, sel_k, sel_weq, anynan;
    double vmin[4];
};

// dynamic LDS: win[W] (float64; dead after the run values: the scratch aliases it) | val[R] | mult[R] (16-bit)
__host__ __device__ inline size_t gv_lds_bytes(int W, int R) {
    size_t w = (size_t)W * 8;
    if (w < sizeof(GvScratch)) w = sizeof(GvScratch);
    w = (w + 15) / 16 * 16;
    return w + (size_t)R * 8 + ((size_t)R * 2 + 15) / 16 * 16;
}

__global__ void __launch_bounds__(kGvThreads) k_gene_fused(
    const double* __restrict__ win, int64_t ldw, int64_t n_rows, int W, const int32_t* __restrict__ run_j0,
    const int32_t* __restrict__ run_cnt, const int32_t* __restrict__ run_mult, int R, int n_cov,
    const int32_t* __restrict__ col_run, int n_cols, const double* __restrict__ thr, int64_t chunksize,
    int64_t row_phase, double* __restrict__ out, int64_t ldg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    size_t woff = (size_t)W * 8;
    if (woff < sizeof(GvScratch)) woff = sizeof(GvScratch);
    woff = (woff + 15) / 16 * 16;
    double* lwin = reinterpret_cast<double*>(gsm);
    GvScratch* sc = reinterpret_cast<GvScratch*>(gsm);  // aliases the windows (used after they are dead)
    double* val = reinterpret_cast<double*>(gsm + woff);
    unsigned short* mult = reinterpret_cast<unsigned short*>(gsm + woff + (size_t)R * 8);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int k1 = (n_cov - 1) / 2, k2 = n_cov / 2;
    for (int i = t; i < R; i += kGvThreads) mult[i] = (unsigned short)run_mult[i];
    const bool vec2 = (ldg % 2 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const double nan = __builtin_nan("");

    for (int64_t cell = blockIdx.x; cell < n_rows; cell += gridDim.x) {
        __syncthreads();  // the previous cell's output phase has read val[]; its scratch use is over
        // ---- 1. the cell's windows
        const double* wr = win + cell * ldw;
        for (int j = t; j < W; j += kGvThreads) lwin[j] = wr[j];
        __syncthreads();
        // ---- 2. run values (np.mean of the covering windows: numpy's pairwise sum / count)
        int nanl = 0;
        for (int r = t; r < R; r += kGvThreads) {
            const int cnt = run_cnt[r];
            const double v = numpy_sum(lwin + run_j0[r], cnt) / (double)cnt;
            val[r] = v;
            nanl |= (v != v);
        }
        __syncthreads();  // windows dead: the scratch may be written
        if (t == 0) sc->anynan = 0;
        __syncthreads();
        if (nanl) sc->anynan = 1;  // benign race
        // ---- 3. weighted median: radix select of rank k1 on the ordered keys, most significant byte first
        unsigned long long prefix = 0ull;
        int k = k1, weq = 0;
        for (int pass = 7; pass >= 0; --pass) {
            sc->hist[t] = 0;
            __syncthreads();
            const int sh = pass * 8;
            for (int r = t; r < R; r += kGvThreads) {
                const unsigned long long key = ordered_key(val[r]);
                const bool in = pass == 7 || ((key ^ prefix) >> (sh + 8)) == 0ull;
                if (in) atomicAdd(&sc->hist[(int)((key >> sh) & 255ull)], (int)mult[r]);
            }
            __syncthreads();
            if (wave == 0) {
                const int4 h = reinterpret_cast<const int4*>(sc->hist)[lane];
                const int c = (h.x + h.y) + (h.z + h.w);
                const int incl = wave_scan_dpp(c);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(incl > k);
                const int L = m ? (int)__builtin_ctzll(m) : 63;
                if (lane == L) {
                    int run = incl - c, d = 3, kk = 0, we = h.w;
                    if (k < run + h.x) { d = 0; kk = k - run; we = h.x; }
                    else if (k < run + h.x + h.y) { d = 1; kk = k - run - h.x; we = h.y; }
                    else if (k < run + h.x + h.y + h.z) { d = 2; kk = k - run - h.x - h.y; we = h.z; }
                    else { kk = k - run - h.x - h.y - h.z; }
                    sc->sel_digit = 4 * L + d;
                    sc->sel_k = kk;
                    sc->sel_weq = we;
                }
            }
            __syncthreads();
            prefix |= (unsigned long long)(unsigned)sc->sel_digit << sh;
            k = sc->sel_k;
            weq = sc->sel_weq;
            // (no third barrier: the next pass writes hist[] -- read by wavefront 0 before the barrier above -- and
            // sel_* only after its own second barrier)
        }
        const double v1 = from_ordered_key(prefix);
        double v2 = v1;
        const bool anynan = sc->anynan != 0;
        if (k2 != k1 && !(k + 1 < weq)) {
            // rank k2 is the smallest value above v1
            double mn = __builtin_inf();
            for (int r = t; r < R; r += kGvThreads) {
                const double v = val[r];
                if (ordered_key(v) > prefix && v < mn) mn = v;
            }
            mn = wave_min_dpp(mn);
            if (lane == 0) sc->vmin[wave] = mn;
            __syncthreads();
            v2 = sc->vmin[0];
            for (int i = 1; i < 4; ++i) v2 = sc->vmin[i] < v2 ? sc->vmin[i] : v2;
        }
        const double med = anynan ? nan : ((k1 == k2) ? v1 : (v1 + v2) / 2.0);
        // ---- 4. the output row, input-column order: value - median, noise filter, NaN where there is no value
        const bool has_thr = thr != nullptr;
        const double th = has_thr ? thr[(cell + row_phase) / chunksize] : 0.0;
        double* orow = out + cell * ldg;
        if (vec2) {
            typedef double f64x2_t __attribute__((ext_vector_type(2)));
            const int n2 = n_cols & ~1;
            for (int c = 2 * t; c < n2; c += 2 * kGvThreads) {
                const int2 rr = *reinterpret_cast<const int2*>(col_run + c);
                double a = rr.x >= 0 ? val[rr.x] - med : nan;
                double b = rr.y >= 0 ? val[rr.y] - med : nan;
                if (has_thr) {
                    if (fabs(a) < th) a = 0.0;
                    if (fabs(b) < th) b = 0.0;
                }
                const f64x2_t q = {a, b};
                __builtin_nontemporal_store(q, reinterpret_cast<f64x2_t*>(orow + c));
            }
            if ((n_cols & 1) && t == 0) {
                const int rx = col_run[n_cols - 1];
                double a = rx >= 0 ? val[rx] - med : nan;
                if (has_thr && fabs(a) < th) a = 0.0;
                orow[n_cols - 1] = a;
            }
        } else {
            for (int c = t; c < n_cols; c += kGvThreads) {
                const int rx = col_run[c];
                double a = rx >= 0 ? val[rx] - med : nan;
                if (has_thr && fabs(a) < th) a = 0.0;
                orow[c] = a;
            }
        }
    }
}

}  // namespace icv

// CDNA4 (gfx950) kernels of the CNV-inference hot path.  Wavefront = 64 lanes.
//
// k_smooth: one 512-thread workgroup (8 wavefronts) owns one cell at a time and walks the
// cells of its shard persistently.  Per cell:
//   L  coalesced 16-byte HBM loads of the cell's expression row (input column order);
//      centre on the reference + clip in the matrix dtype; scatter through the plan's
//      column->position table into the LDS row (chromosome-sorted, block-padded order)
//   S  per block of B consecutive genes: S0 = sum v, S1 = sum r*v (float64, registers),
//      then stored over the (dead) row                                   [B > 1 only]
//   W  every window = sum over its blocks of (a_m*S0 +- S1) / sum(weights)  (pyramid),
//      or sum S0 / G_c (flat small-chromosome window); float64, into LDS
//   M  per-cell median over all W windows: value-space bisection on the LDS-resident
//      windows with workgroup-wide counts, exact rank resolution of the last <= 64
//      candidates
//   O  x_res = window - median -> float32 coalesced store; per-cell sum / sum of squares
// The path is HBM-bound (reads 4*G + writes 4*W bytes per cell); no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace icv {

constexpr int NT = 512;
constexpr int NWAVE = NT / 64;

struct KParams {
    // input matrix
    const void* values;
    const int64_t* indptr;
    const int32_t* indices;
    int64_t n_rows;
    int64_t ld;
    int32_t n_cols;
    int32_t vec_ok;  // dense: 16-byte vector loads allowed (alignment + ld)
    // reference, matrix dtype, input column order; bounded == 0: ref_hi unused
    const void* ref_lo;
    const void* ref_hi;
    const void* zrow;  // CSR: padded row of centre_clip(0) values (matrix dtype)
    int64_t zrow_bytes;
    int32_t bounded;
    int32_t trunc;
    double cap;
    // plan tables (device)
    const int32_t* dst;  // n_cols: padded position or -1
    const int32_t* src;  // Gp: input column or -1
    const int32_t* w_start;
    const int32_t* w_len;
    const double* w_denom;
    const uint16_t* dst16;   // fast path: padded table of LDS positions, Gp (trash slot) = masked
    const int32_t* w_pack;   // ws path: per window (start block & 0xffff) | (len << 16)
    const uint16_t* pos16;   // ws CSR path: per stored entry, LDS position (k_csr_prepare)
    const float* cvals;      //              and centred + clipped value
    const int32_t* pad_idx;  // padded positions that hold no gene (must read as 0)
    int32_t n_pad;
    int32_t _pad0;
    double pyr_den;          // sum of the pyramid weights, and its correctly rounded reciprocal
    double pyr_rcp;
    double med_bound;        // |window| <= med_bound (from the clip value); fast path bracket
    int32_t B, NB, Gp, W;
    int32_t win_off, scratch_off, hist_off, _pad1;
    // outputs
    float* out;
    int64_t ldo;
    double* cell_median;
    double* cell_stats;
    unsigned long long* dbg;  // developer diagnostic: per-phase shader-cycle totals (ICV_PHASE_PROFILE=1)
    // k_smooth only: process rows row_list[0 .. *row_count) instead of 0 .. n_rows (cells a fast
    // kernel handed back); k_smooth_ws appends to the same list
    int64_t* row_list;
    int* row_count;
    double* win_out;  // k_smooth only (optional): float64 smoothed windows before centring, n_rows x W
    // k_smooth_ws / k_smooth_sp: per-wavefront partial moments, n_rows x 8 x {sum, sum of squares}; lane 0 of every
    // wavefront stores its pair straight to HBM (no LDS round trip, no single-thread reduction on the per-cell
    // critical path) and k_stats_finish adds the eight pairs in a fixed order
    double* cell_part;
    // k_smooth, Layout::win_global: gridDim.x lines of W float64 windows in HBM instead of the LDS window array
    double* win_scratch;
    // k_smooth, chromosome-group passes of a gene set whose row does not fit LDS (icv_api.hip: smooth_split):
    // win_only != 0: write the float64 windows to win_out[cell * win_ld + j] and stop there (no median, no x_res)
    int64_t win_ld;
    int32_t win_only, _pad3;
    // k_smooth_x16<CHUNK>: moments per noise-threshold chunk instead of per cell.  Chunk k holds the rows r with
    // (r + row_phase) / chunksize == k; chunk_part[(k * gridDim.x + workgroup) * 16 + wavefront] = {sum, sum of
    // squares} of that wavefront's windows over the workgroup's cells of the chunk (zero-filled by the host)
    int64_t chunksize, row_phase;
    double* chunk_part;
    // k_smooth_x16: per thread, first block (bits 0-11) and first window (12-23) of its two adjacent windows,
    // window valid (24, 25), full pyramid window (26, 27); then per thread: 0, or for a flat first window its
    // block count | gene count << 16
    const uint32_t* x16_wdesc;
    int32_t x16_half, _pad4;  // k_smooth_x16: slots of the even-block {S0,S1} array
    // k_smooth_se (CSR float32, stored entries only).  Per block the gene offset of its first gene inside its
    // chromosome; per input column {LDS address of the block's bins, ref_lo, in-block offset * 2^k1, clip(0 - ref)}
    // (k_se_table), ref_hi per column (bounded references); per window {w0, w1, zero-row sum} (k_se_wtab); per block
    // its first-gene offset * 2^(k1-k0); the fixed-point scales 2^k0, 2^-k0, 2^-k1, 2^(k1-k0)
    const int32_t* blk_g0;
    const void* sd_tab;
    const float* sd_tab_hi;
    const void* sd_wtab;
    const double* sd_g16;
    double sd_scale, sd_qinv, sd_q1inv, sd_r;
    int32_t sd_window, _pad5;
};

struct Scratch {
    double dred[3][2][NWAVE];  // [stage: 0 min/max, 1 second-middle, 2 moments][slot][wave]
    int ired[2][NWAVE];
    int ncand;
    int flag;
    double cand[64];
    double sel;
};
static_assert(sizeof(Scratch) <= 1024, "Scratch must fit kScratchBytes");

// ---------------------------------------------------------------------------------------
// element-wise step 1 + 2 (reference :422-436), in the matrix dtype T
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T centre_clip(T x, T lo, T hi, T cap, int bounded, int trunc_int) {
    T v;
    if (!bounded) {
        v = x - lo;
    } else {
        const T above = x - hi, below = x - lo;
        v = (x > hi) ? above : ((x < lo) ? below : T(0));
        if (trunc_int & 1) v = (T)trunc((double)v);
        if (trunc_int & 2) v = (T)(float)v;
    }
    v = v < -cap ? -cap : v;  // np.clip == minimum(maximum(v, -cap), cap); NaN stays NaN
    v = v > cap ? cap : v;
    return v;
}

// ---------------------------------------------------------------------------------------
// canonical float64 evaluation order of one window.  Used by the smoothing kernel (values
// from LDS) and by the exact tie-break of the threshold kernel (values re-read from HBM):
// both must produce bit-identical results.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void block_accumulate(double v, int r, double& s0, double& s1) {
    s0 = s0 + v;
    s1 = fma((double)r, v, s1);
}

// acc / denom.  Pyramid windows share one denominator: q = RN(acc * RN(1/d)) corrected by one
// residual step is the correctly rounded quotient (Markstein) at 3 FMAs instead of a ~40
// instruction IEEE division; flat windows (one per small chromosome) use the plain division.
__device__ __forceinline__ double finish_window(double acc, int len, double pyr_den, double pyr_rcp,
                                                double flat_den) {
    if (len > 0) {
        const double q = acc * pyr_rcp;
        const double r = fma(-q, pyr_den, acc);
        return fma(r, pyr_rcp, q);
    }
    return acc / flat_den;
}

template <typename F>  // F(m, &S0, &S1): partial sums of block m of the window
__device__ __forceinline__ double window_from_blocks(int len, int B, F blk) {
    double acc = 0.0;
    if (len > 0) {
        const int nb = len / B, hb = nb / 2;
#pragma unroll 2
        for (int m = 0; m < nb; ++m) {
            double S0, S1;
            blk(m, S0, S1);
            if (m < hb) {
                acc = fma((double)(m * B + 1), S0, acc);
                acc = acc + S1;
            } else {
                acc = fma((double)(len - m * B), S0, acc);
                acc = acc - S1;
            }
        }
    } else {
        const int nb = (-len) / B;
#pragma unroll 1
        for (int m = 0; m < nb; ++m) {
            double S0, S1;
            blk(m, S0, S1);
            acc = acc + S0;
        }
    }
    return acc;
}

template <typename F>  // F(k): clipped value k of the window as double
__device__ __forceinline__ double window_direct(int len, F val) {
    double acc = 0.0;
    if (len > 0) {
        for (int k = 0; k < len; ++k) {
            int w = (k + 1 < len - k) ? (k + 1) : (len - k);
            acc = fma((double)w, val(k), acc);
        }
    } else {
        for (int k = 0; k < -len; ++k) acc = acc + val(k);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------
// wavefront / workgroup reductions (64-lane)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// workgroup-wide sum of an int; `par` alternates between consecutive calls (no WAR hazard)
__device__ __forceinline__ int wg_sum_i(int v, Scratch* sc, int par) {
    v = wave_sum_i(v);
    if ((threadIdx.x & 63) == 0) sc->ired[par][threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) s += sc->ired[par][i];
    return s;
}

__device__ __forceinline__ double prev_double(double x) {
    // largest double strictly below finite x
    if (x == 0.0) return -4.9406564584124654e-324;
    long long b = __double_as_longlong(x);
    b += (x > 0.0) ? -1 : 1;
    return __longlong_as_double(b);
}
__device__ __forceinline__ unsigned long long ordered_key(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_key(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// ---------------------------------------------------------------------------------------
// DPP wavefront reductions (no LDS traffic, unlike __shfl): butterfly inside each row of 16
// lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row values are
// read back as scalars and combined in a fixed order -> deterministic, result uniform.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_move_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror
    return ((readlane_d(v, 0) + readlane_d(v, 16)) + readlane_d(v, 32)) + readlane_d(v, 48);
}
// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (no LDS traffic)
__device__ __forceinline__ int wave_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ double wave_min_dpp(double v) {
    double o;
    o = dpp_move<0xB1>(v); v = o < v ? o : v;
    o = dpp_move<0x4E>(v); v = o < v ? o : v;
    o = dpp_move<0x141>(v); v = o < v ? o : v;
    o = dpp_move<0x140>(v); v = o < v ? o : v;
    double a = readlane_d(v, 0), b = readlane_d(v, 16), c = readlane_d(v, 32), d = readlane_d(v, 48);
    a = b < a ? b : a;
    c = d < c ? d : c;
    return c < a ? c : a;
}
__device__ __forceinline__ double wave_max_dpp(double v) { return -wave_min_dpp(-v); }

// ---------------------------------------------------------------------------------------
// the fused smoothing kernel
// ---------------------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; using itype = int4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; using itype = int2; static constexpr int N = 2; };

template <typename T>
__device__ __forceinline__ T vget(const typename Vec16<T>::type& v, int i);
template <> __device__ __forceinline__ float vget<float>(const float4& v, int i) {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}
template <> __device__ __forceinline__ double vget<double>(const double2& v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ int iget(const int4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ int iget(const int2& v, int i) { return i == 0 ? v.x : v.y; }

template <typename T, bool CSR, int MAXB /* 0 = direct form */>
__global__ void __launch_bounds__(NT) k_smooth(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* row = reinterpret_cast<T*>(smem);
    double* S01 = reinterpret_cast<double*>(smem);
    double* win = P.win_scratch ? P.win_scratch + (int64_t)blockIdx.x * P.W
                                : reinterpret_cast<double*>(smem + P.win_off);
    Scratch* sc = reinterpret_cast<Scratch*>(smem + P.scratch_off);

    using V = typename Vec16<T>::type;
    using IV = typename Vec16<T>::itype;
    constexpr int VN = Vec16<T>::N;
    constexpr int U = 4;  // 16-byte loads kept in flight per lane per batch

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const T cap = (T)P.cap;
    const T* ref_lo = static_cast<const T*>(P.ref_lo);
    const T* ref_hi = P.bounded ? static_cast<const T*>(P.ref_hi) : ref_lo;
    const int W = P.W, B = P.B, NB = P.NB;

    // pad slots of the row are never written by the scatter: zero them once (direct form, no
    // aliasing) or after every cell (blocked form, the aliased S01/win overwrite them).
    for (int i = t; i < P.n_pad; i += NT) row[P.pad_idx[i]] = T(0);
    __syncthreads();

    unsigned long long tlast = 0, tacc[5] = {0, 0, 0, 0, 0};
#define ICV_PHASE(i)                                              \
    if (P.dbg && t == 0) {                                        \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        tacc[i] += now_ - tlast;                                  \
        tlast = now_;                                             \
    }
    if (P.dbg && t == 0) tlast = __builtin_amdgcn_s_memtime();

    const int64_t n_work = P.row_list ? (int64_t)*P.row_count : P.n_rows;
    for (int64_t work = blockIdx.x; work < n_work; work += gridDim.x) {
        const int64_t cell = P.row_list ? P.row_list[work] : work;
        // ---------------- L: load, centre, clip, scatter --------------------------------
        if constexpr (!CSR) {
            const T* xrow = static_cast<const T*>(P.values) + cell * P.ld;
            int done = 0;
            if (P.vec_ok) {
                const int nvec = P.n_cols / VN;
                const V* xv = reinterpret_cast<const V*>(xrow);
                const IV* dv = reinterpret_cast<const IV*>(P.dst);
                const V* lov = reinterpret_cast<const V*>(ref_lo);
                const V* hiv = reinterpret_cast<const V*>(ref_hi);
                for (int base = 0; base < nvec; base += NT * U) {
                    V x[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        int i = base + u * NT + t;
                        if (i < nvec) x[u] = xv[i];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        int i = base + u * NT + t;
                        if (i < nvec) {
                            IV d = dv[i];
                            V lo = lov[i];
                            V hi = hiv[i];
#pragma unroll
                            for (int e = 0; e < VN; ++e) {
                                int q = iget(d, e);
                                if (q >= 0)
                                    row[q] = centre_clip<T>(vget<T>(x[u], e), vget<T>(lo, e), vget<T>(hi, e), cap,
                                                            P.bounded, P.trunc);
                            }
                        }
                    }
                }
                done = nvec * VN;
            }
            for (int g = done + t; g < P.n_cols; g += NT) {
                int q = P.dst[g];
                if (q >= 0) row[q] = centre_clip<T>(xrow[g], ref_lo[g], ref_hi[g], cap, P.bounded, P.trunc);
            }
        } else {
            // implicit zeros are not zero after centring: start from clip(centre(0)) ...
            const V* zv = reinterpret_cast<const V*>(P.zrow);
            V* rv = reinterpret_cast<V*>(row);
            const int nvec = (P.Gp + VN - 1) / VN;
            for (int i = t; i < nvec; i += NT) rv[i] = zv[i];
            __syncthreads();
            // ... then overwrite the stored entries
            const int64_t s = P.indptr[cell], e = P.indptr[cell + 1];
            const T* vals = static_cast<const T*>(P.values);
            for (int64_t k = s + t; k < e; k += NT) {
                int g = P.indices[k];
                int q = P.dst[g];
                if (q >= 0) row[q] = centre_clip<T>(vals[k], ref_lo[g], ref_hi[g], cap, P.bounded, P.trunc);
            }
        }
        __syncthreads();
        ICV_PHASE(0)

        // ---------------- S + W: block partial sums, windows ----------------------------
        double lmin = __builtin_inf(), lmax = -__builtin_inf();
        int lnan = 0;
        if constexpr (MAXB > 0) {
            double s0[MAXB], s1[MAXB];
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                s0[i] = 0.0;
                s1[i] = 0.0;
                const int b = t + i * NT;
                if (b < NB) {
                    const T* rp = row + b * B;
                    for (int r = 0; r < B; ++r) block_accumulate((double)rp[r], r, s0[i], s1[i]);
                }
            }
            __syncthreads();  // every read of the row is done: alias it
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int b = t + i * NT;
                if (b < NB) {
                    S01[2 * b] = s0[i];
                    S01[2 * b + 1] = s1[i];
                }
            }
            __syncthreads();
            ICV_PHASE(1)
            for (int j = t; j < W; j += NT) {
                const int st = P.w_start[j], ln = P.w_len[j];
                const double* sp = S01 + 2 * (st / B);
                double v = window_from_blocks(ln, B, [&](int m, double& a, double& b2) {
                    a = sp[2 * m];
                    b2 = sp[2 * m + 1];
                });
                v = finish_window(v, ln, P.pyr_den, P.pyr_rcp, P.w_denom[j]);
                if (P.win_only) P.win_out[cell * P.win_ld + j] = v;
                else win[j] = v;
                lmin = v < lmin ? v : lmin;
                lmax = v > lmax ? v : lmax;
                lnan |= (v != v);
            }
        } else {
            for (int j = t; j < W; j += NT) {
                const int st = P.w_start[j], ln = P.w_len[j];
                const T* rp = row + st;
                double v = window_direct(ln, [&](int k) { return (double)rp[k]; });
                v = finish_window(v, ln, P.pyr_den, P.pyr_rcp, P.w_denom[j]);
                if (P.win_only) P.win_out[cell * P.win_ld + j] = v;
                else win[j] = v;
                lmin = v < lmin ? v : lmin;
                lmax = v > lmax ? v : lmax;
                lnan |= (v != v);
            }
        }
        if (P.win_only) {  // one chromosome group of a split gene set: the windows are all this pass produces
            if constexpr (MAXB > 0) {
                __syncthreads();
                for (int i = t; i < P.n_pad; i += NT) row[P.pad_idx[i]] = T(0);
            }
            __syncthreads();
            continue;
        }
        lmin = wave_min(lmin);
        lmax = wave_max(lmax);
        if (lane == 0) {
            sc->dred[0][0][wave] = lmin;
            sc->dred[0][1][wave] = lmax;
        }
        const int anynan = __syncthreads_or(lnan);  // also publishes win[] and dred
        ICV_PHASE(2)

        // ---------------- M: median over all W windows (np.median, reference :442) ------
        double med;
        if (anynan) {
            med = __builtin_nan("");
        } else {
            double gmin = sc->dred[0][0][0], gmax = sc->dred[0][1][0];
#pragma unroll
            for (int i = 1; i < NWAVE; ++i) {
                double a = sc->dred[0][0][i], b = sc->dred[0][1][i];
                gmin = a < gmin ? a : gmin;
                gmax = b > gmax ? b : gmax;
            }
            // order statistic k1 (0-based) lies in (lo, hi]; cnt_x = #{win <= x}
            double lo = prev_double(gmin), hi = gmax;
            int cnt_lo = 0, cnt_hi = W;
            const int k1 = (W - 1) / 2;
            int it = 0;
            while (cnt_hi - cnt_lo > 64 && it < 120) {
                double mid;
                if (it < 48) {
                    mid = 0.5 * lo + 0.5 * hi;
                } else {
                    unsigned long long a = ordered_key(lo), b = ordered_key(hi);
                    mid = from_ordered_key(a + ((b - a) >> 1));
                }
                if (!(mid > lo && mid < hi)) break;  // adjacent doubles: every candidate == hi
                int c = 0;
                for (int j = t; j < W; j += NT) c += (win[j] <= mid) ? 1 : 0;
                c = wg_sum_i(c, sc, it & 1);
                if (c > k1) { hi = mid; cnt_hi = c; } else { lo = mid; cnt_lo = c; }
                ++it;
            }
            double a;
            if (cnt_hi - cnt_lo <= 64) {
                if (t == 0) sc->ncand = 0;
                __syncthreads();
                for (int j = t; j < W; j += NT) {
                    double x = win[j];
                    if (x > lo && x <= hi) {
                        int idx = atomicAdd(&sc->ncand, 1);
                        if (idx < 64) sc->cand[idx] = x;
                    }
                }
                __syncthreads();
                if (t < 64) {
                    const int n = sc->ncand < 64 ? sc->ncand : 64;
                    const double mine = (t < n) ? sc->cand[t] : 0.0;
                    int rank = 0;
                    for (int i = 0; i < n; ++i) {
                        double o = sc->cand[i];
                        rank += (o < mine || (o == mine && i < t)) ? 1 : 0;
                    }
                    if (t < n && rank == k1 - cnt_lo) sc->sel = mine;
                }
                __syncthreads();
                a = sc->sel;
            } else {
                a = hi;
            }
            med = a;
            if ((W & 1) == 0) {
                // second middle element: a again if it is repeated, else the smallest value above a
                __syncthreads();  // ired[0] may still be read by the last bisection step
                int c = 0;
                double mn = __builtin_inf();
                for (int j = t; j < W; j += NT) {
                    double x = win[j];
                    c += (x <= a) ? 1 : 0;
                    if (x > a) mn = x < mn ? x : mn;
                }
                mn = wave_min(mn);
                c = wave_sum_i(c);
                if (lane == 0) {
                    sc->dred[1][0][wave] = mn;
                    sc->ired[0][wave] = c;
                }
                __syncthreads();
                int ctot = 0;
                double mtot = __builtin_inf();
#pragma unroll
                for (int i = 0; i < NWAVE; ++i) {
                    ctot += sc->ired[0][i];
                    double m2 = sc->dred[1][0][i];
                    mtot = m2 < mtot ? m2 : mtot;
                }
                const double b = (ctot > k1 + 1) ? a : mtot;
                med = (a + b) / 2.0;
            }
        }

        ICV_PHASE(3)
        // ---------------- O: centre, store, per-cell moments ----------------------------
        double sum = 0.0, sq = 0.0;
        float* orow = P.out + cell * P.ldo;
        for (int j = t; j < W; j += NT) {
            const double y = win[j] - med;
            orow[j] = (float)y;
            if (P.win_out) P.win_out[cell * P.win_ld + j] = win[j];
            sum = sum + y;
            sq = fma(y, y, sq);
        }
        sum = wave_sum_dpp(sum);  // same reduction tree as k_smooth_fast: bit-identical moments
        sq = wave_sum_dpp(sq);
        if (lane == 0) {
            sc->dred[2][0][wave] = sum;
            sc->dred[2][1][wave] = sq;
        }
        __syncthreads();
        if (t == 0) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < NWAVE; ++i) {
                s += sc->dred[2][0][i];
                q += sc->dred[2][1][i];
            }
            P.cell_stats[2 * cell] = s;
            P.cell_stats[2 * cell + 1] = q;
            P.cell_median[cell] = med;
        }
        // re-zero the pad slots clobbered by the aliased S01 / win arrays
        if constexpr (MAXB > 0) {
            __syncthreads();
            for (int i = t; i < P.n_pad; i += NT) row[P.pad_idx[i]] = T(0);
        }
        __syncthreads();
        ICV_PHASE(4)
    }
    if (P.dbg && t == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(P.dbg + i, tacc[i]);
#undef ICV_PHASE
}

// ---------------------------------------------------------------------------------------
// raw buffer descriptors (out-of-range lanes read 0, no traffic) for the register-prefetch kernels
// (icv_kernel_ws.hpp, icv_kernel_x16.hpp)
// ---------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    // raw buffer descriptor: base, stride 0, num_records = bytes (out-of-range lanes read 0)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------
// CSR: padded row of centre_clip(0, ref) (implicit zeros after centring, reference :423)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void k_zero_row(KParams P, T* zrow, int n_alloc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_alloc) return;
    T v = T(0);
    if (i < P.Gp) {
        int g = P.src[i];
        if (g >= 0) {
            const T* lo = static_cast<const T*>(P.ref_lo);
            const T* hi = P.bounded ? static_cast<const T*>(P.ref_hi) : lo;
            v = centre_clip<T>(T(0), lo[g], hi[g], (T)P.cap, P.bounded, P.trunc);
        }
    }
    zrow[i] = v;
}

// CSR fast path: every stored entry -> {LDS position of its column, centred + clipped value}.
// Element-wise over the entries (centring depends on the column only), coalesced.
__global__ void __launch_bounds__(256) k_csr_prepare(const KParams P, int64_t k0, int64_t k1, uint16_t* pos16,
                                                     float* cvals) {
    const float* lo = static_cast<const float*>(P.ref_lo);
    const float* hi = P.bounded ? static_cast<const float*>(P.ref_hi) : lo;
    const float* vals = static_cast<const float*>(P.values);
    for (int64_t k = k0 + (int64_t)blockIdx.x * 256 + threadIdx.x; k < k1; k += (int64_t)gridDim.x * 256) {
        const int g = P.indices[k];
        pos16[k] = P.dst16[g];
        cvals[k] = centre_clip<float>(vals[k], lo[g], hi[g], (float)P.cap, P.bounded, P.trunc);
    }
}

// ---------------------------------------------------------------------------------------
// step 5a: per-chunk threshold = dynamic_threshold * population std  (reference :450)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_chunk_thr(const double* stats, int64_t n_rows, int64_t chunksize,
                                                   int64_t row_phase, int n_windows, double dyn, double* thr) {
    __shared__ double ss[256], sq[256];
    const int64_t k = blockIdx.x;
    int64_t r0 = k * chunksize - row_phase, r1 = r0 + chunksize;
    if (r0 < 0) r0 = 0;
    if (r1 > n_rows) r1 = n_rows;
    double s = 0.0, q = 0.0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        s += stats[2 * r];
        q += stats[2 * r + 1];
    }
    ss[threadIdx.x] = s;
    sq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            ss[threadIdx.x] += ss[threadIdx.x + o];
            sq[threadIdx.x] += sq[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)(r1 - r0) * (double)n_windows;
        const double mean = ss[0] / n;
        double var = (sq[0] - ss[0] * mean) / n;
        if (var < 0.0) var = 0.0;
        thr[k] = dyn * sqrt(var);
    }
}

// the same from the per-(workgroup, wavefront) chunk partials of k_smooth_x16<CHUNK>; hb_stats: per-row moments,
// zero except for the rows the generic kernel recomputed (cells handed back by k_smooth_x16).  Fixed order.
__global__ void __launch_bounds__(256) k_chunk_thr_part(const double* chunk_part, int n_part, const double* hb_stats,
                                                        int64_t n_rows, int64_t chunksize, int64_t row_phase,
                                                        int n_windows, double dyn, double* thr) {
    __shared__ double ss[256], sq[256];
    const int64_t k = blockIdx.x;
    int64_t r0 = k * chunksize - row_phase, r1 = r0 + chunksize;
    if (r0 < 0) r0 = 0;
    if (r1 > n_rows) r1 = n_rows;
    double s = 0.0, q = 0.0;
    const double2* part = reinterpret_cast<const double2*>(chunk_part) + k * n_part;
    for (int i = threadIdx.x; i < n_part; i += 256) {
        const double2 v = part[i];
        s += v.x;
        q += v.y;
    }
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        s += hb_stats[2 * r];
        q += hb_stats[2 * r + 1];
    }
    ss[threadIdx.x] = s;
    sq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            ss[threadIdx.x] += ss[threadIdx.x + o];
            sq[threadIdx.x] += sq[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)(r1 - r0) * (double)n_windows;
        const double mean = ss[0] / n;
        double var = (sq[0] - ss[0] * mean) / n;
        if (var < 0.0) var = 0.0;
        thr[k] = dyn * sqrt(var);
    }
}

// ---------------------------------------------------------------------------------------
// step 5b: zero |x| < thr, decided in float64 (reference :451).  The stored float32 value
// decides except when it lies within one float32 ulp of float32(thr): then the window is recomputed
// from the input with the canonical evaluation order above (bit-identical to k_smooth) and compared
// in float64.  The one-ulp band makes the decision independent of the smoothing kernel that wrote
// x_res: k_smooth_se sums its windows in another float64 order (equal to ~1e-12), so its float32
// value can differ from the canonical one in the last bit.
// ---------------------------------------------------------------------------------------
// |y| against the float32 threshold: -1 below, +1 above, 0 float32 cannot decide (a NaN value, or a threshold that is
// NaN, zero or negative -- nothing is below those: +1)
__device__ __forceinline__ int thr_compare(float a, float thf) {
    if (!(thf > 0.0f) || !(a == a)) return 1;
    const int d = __float_as_int(a) - __float_as_int(thf);  // both non-negative: the bit patterns are ordered
    return d < -1 ? -1 : (d > 1 ? 1 : 0);
}

// clipped, centred value of padded position pp of one cell, re-read from the input matrix
template <typename T, bool CSR>
__device__ double value_at(const KParams& P, int64_t cell, int pp) {
    const int g = P.src[pp];
    if (g < 0) return 0.0;
    const T* ref_lo = static_cast<const T*>(P.ref_lo);
    const T* ref_hi = P.bounded ? static_cast<const T*>(P.ref_hi) : ref_lo;
    T x = T(0);
    if constexpr (!CSR) {
        x = static_cast<const T*>(P.values)[cell * P.ld + g];
    } else {
        // column indices are sorted within a row (the host driver sorts them): binary search
        int64_t lo = P.indptr[cell], hi = P.indptr[cell + 1];
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (P.indices[mid] < g) lo = mid + 1;
            else hi = mid;
        }
        if (lo < P.indptr[cell + 1] && P.indices[lo] == g) x = static_cast<const T*>(P.values)[lo];
    }
    return (double)centre_clip<T>(x, ref_lo[g], ref_hi[g], (T)P.cap, P.bounded, P.trunc);
}

// canonical (bit-identical to k_smooth) float64 window from a value accessor val(k), k = 0 .. |len|-1
template <typename F>
__device__ double window_canonical(const KParams& P, int j, F val) {
    const int ln = P.w_len[j];
    double acc;
    if (P.B > 1) {
        const int B = P.B;
        acc = window_from_blocks(ln, B, [&](int m, double& s0, double& s1) {
            s0 = 0.0;
            s1 = 0.0;
            for (int r = 0; r < B; ++r) block_accumulate(val(m * B + r), r, s0, s1);
        });
    } else {
        acc = window_direct(ln, val);
    }
    return finish_window(acc, ln, P.pyr_den, P.pyr_rcp, P.w_denom[j]);
}

constexpr int kTieBuf = 2048;  // genes of a tied window staged in LDS (longer windows: serial path)

template <typename T, bool CSR>
__global__ void __launch_bounds__(256) k_apply_thr(const KParams P, const double* thr, int64_t chunksize,
                                                   int64_t row_phase) {
    __shared__ int tie_n;
    __shared__ int tie_j[32];
    __shared__ double vals[kTieBuf];
    const int64_t cell = blockIdx.x;
    const double th = thr[(cell + row_phase) / chunksize];
    const float thf = (float)th;
    float* orow = P.out + cell * P.ldo;
    if (threadIdx.x == 0) tie_n = 0;
    __syncthreads();
    // returns the value to store for window j (0 below the threshold); ties are queued / resolved exactly
    auto decide = [&](int j, float y) -> float {
        if (y == 0.0f) return y;  // zero either way
        const float a = fabsf(y);
        const int cmp = thr_compare(a, thf);
        if (cmp < 0) return 0.0f;
        if (cmp == 0) {
            // float32 cannot decide: queue the window for an exact float64 recomputation
            const int idx = atomicAdd(&tie_n, 1);
            if (idx < 32) {
                tie_j[idx] = j;
            } else {  // > 32 ties in one row: resolve serially
                const int st = P.w_start[j];
                const double yd = window_canonical(P, j, [&](int k) { return value_at<T, CSR>(P, cell, st + k); }) -
                                  P.cell_median[cell];
                if (fabs(yd) < th) return 0.0f;
            }
        }
        return y;
    };
    if ((reinterpret_cast<uintptr_t>(orow) & 7) == 0) {  // 8-byte aligned row: two windows per load / store
        float2* o2 = reinterpret_cast<float2*>(orow);
        const int half = P.W >> 1;
        // batches of four pairs per thread: all loads of a batch are in flight before the first store
        for (int q0 = threadIdx.x; q0 < half; q0 += 4 * 256) {
            float2 y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) y[u] = (q0 + u * 256 < half) ? o2[q0 + u * 256] : make_float2(0.0f, 0.0f);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * 256;
                if (q < half) {
                    const float2 z = make_float2(decide(2 * q, y[u].x), decide(2 * q + 1, y[u].y));
                    if (z.x != y[u].x || z.y != y[u].y) o2[q] = z;
                }
            }
        }
        if ((P.W & 1) && threadIdx.x == 0) {
            const int j = P.W - 1;
            const float y = orow[j], z = decide(j, y);
            if (z != y) orow[j] = z;
        }
    } else {
        for (int j = threadIdx.x; j < P.W; j += 256) {
            const float y = orow[j], z = decide(j, y);
            if (z != y) orow[j] = z;
        }
    }
    __syncthreads();
    const int nt = tie_n < 32 ? tie_n : 32;
    for (int i = 0; i < nt; ++i) {  // rare (about one window in 1e7): the block recomputes it together
        const int j = tie_j[i];
        const int st = P.w_start[j], ln = P.w_len[j];
        const int len = ln > 0 ? ln : -ln;
        if (len <= kTieBuf) {
            for (int k = threadIdx.x; k < len; k += 256) vals[k] = value_at<T, CSR>(P, cell, st + k);
            __syncthreads();
            if (threadIdx.x == 0) {
                const double yd = window_canonical(P, j, [&](int k) { return vals[k]; }) - P.cell_median[cell];
                if (fabs(yd) < th) orow[j] = 0.0f;
            }
            __syncthreads();
        } else if (threadIdx.x == 0) {
            const double yd = window_canonical(P, j, [&](int k) { return value_at<T, CSR>(P, cell, st + k); }) -
                              P.cell_median[cell];
            if (fabs(yd) < th) orow[j] = 0.0f;
        }
    }
}

// step 5b fused with the CSR count (public path: X_cnv leaves the GPU as CSR, reference :455): the same decision
// as k_apply_thr, but x_res is left untouched -- the kept entries (|x| >= thr decided in float64, x != 0; NaN is
// kept) are recorded as one bit per window and counted per row; k_csr_fill_masked then packs them.  x_res is read
// twice in total and never rewritten.
template <typename T, bool CSR>
__global__ void __launch_bounds__(256) k_thr_mask(const KParams P, const double* thr, int64_t chunksize,
                                                  int64_t row_phase, unsigned long long* mask, int n_words,
                                                  int64_t* row_nnz) {
    __shared__ int tie_n, cnt[4];
    __shared__ int tie_j[32];
    __shared__ double vals[kTieBuf];
    // last rows first: the smoothing kernel has just written x_res, its tail is still in the Infinity Cache (and this
    // pass leaves the HEAD there for k_csr_fill_masked, which walks forward)
    const int64_t cell = (int64_t)gridDim.x - 1 - blockIdx.x;
    const bool has_thr = thr != nullptr;
    const double th = has_thr ? thr[(cell + row_phase) / chunksize] : 0.0;
    const float thf = (float)th;
    const float* orow = P.out + cell * P.ldo;
    unsigned long long* mrow = mask + cell * (int64_t)n_words;
    if (threadIdx.x == 0) tie_n = 0;
    __syncthreads();
    int kept = 0;
    // batches of four values per thread: all loads of a batch are in flight before the first decision
    for (int j0 = 0; j0 < P.W; j0 += 4 * 256) {
        float yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 256 + (int)threadIdx.x;
            yv[u] = j < P.W ? orow[j] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 256 + (int)threadIdx.x;
            bool keep = false;
            if (j < P.W) {
                const float y = yv[u];
                const float a = fabsf(y);
                keep = y != 0.0f;  // NaN: kept (stored explicitly, like csr_matrix(x_res))
                if (has_thr) {
                    const int cmp = thr_compare(a, thf);
                    if (cmp < 0) keep = false;
                    else if (cmp == 0 && keep) {  // float32 cannot decide: exact float64 recomputation below
                        const int idx = atomicAdd(&tie_n, 1);
                        if (idx < 32) {
                            tie_j[idx] = j;
                        } else {  // > 32 ties in one row: resolve serially
                            const int st = P.w_start[j];
                            const double yd = window_canonical(P, j, [&](int k) { return value_at<T, CSR>(P, cell, st + k); }) -
                                              P.cell_median[cell];
                            if (fabs(yd) < th) keep = false;
                        }
                    }
                }
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if ((threadIdx.x & 63) == 0 && j < P.W) mrow[j >> 6] = m;
            kept += (threadIdx.x & 63) == 0 ? __popcll(m) : 0;
        }
    }
    __syncthreads();  // mask words and the tie list are complete (workgroup scope)
    const int nt = tie_n < 32 ? tie_n : 32;
    int dropped = 0;
    for (int i = 0; i < nt; ++i) {  // rare (about one window in 1e7): the block recomputes it together
        const int j = tie_j[i];
        const int st = P.w_start[j], ln = P.w_len[j];
        const int len = ln > 0 ? ln : -ln;
        bool drop = false;
        if (len <= kTieBuf) {
            for (int k = threadIdx.x; k < len; k += 256) vals[k] = value_at<T, CSR>(P, cell, st + k);
            __syncthreads();
            if (threadIdx.x == 0)
                drop = fabs(window_canonical(P, j, [&](int k) { return vals[k]; }) - P.cell_median[cell]) < th;
            __syncthreads();
        } else if (threadIdx.x == 0) {
            drop = fabs(window_canonical(P, j, [&](int k) { return value_at<T, CSR>(P, cell, st + k); }) -
                        P.cell_median[cell]) < th;
        }
        if (threadIdx.x == 0 && drop) {
            mrow[j >> 6] &= ~(1ull << (j & 63));
            ++dropped;
        }
    }
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0) row_nnz[cell] = (int64_t)(cnt[0] + cnt[1] + cnt[2] + cnt[3] - dropped);
}

// Step 5b + `csr_matrix(x_res)` (reference :449-455) in ONE pass over x_res: the decision of k_thr_mask, the rows' kept
// counts, their offsets in the packed output from a decoupled look-back, and the entries (int32 column, float64 value)
// written straight from the rows -- x_res is read from HBM once (the packing re-reads the workgroup's rows from L2),
// no mask array in HBM, no separate scan, no second kernel.
//   A workgroup takes `rpw` consecutive rows per ticket (ticket counter: every predecessor of a waiting workgroup is
//   running or done).  status[t]: 64-bit word per ticket {2-bit flag, 62-bit value}: 1 = the ticket's own
//   count, 2 = the inclusive prefix of tickets 0..t (one relaxed agent-scope 8-byte store: value and flag cannot tear).
//   The look-back examines 64 predecessors per step and waits only for those nearer than the nearest known prefix; a
//   step covers 64 x rpw rows.  `ticket` and `status` are zeroed by the caller.
//   indptr[0 .. n_rows] from 0; indices / data of capacity `cap` entries: an entry beyond `cap` is dropped (the caller
//   sizes for the worst case or checks indptr[n_rows]).  The output does not depend on the order of completion.
constexpr unsigned long long kPackAgg = 1ull << 62, kPackPfx = 2ull << 62, kPackVal = (1ull << 62) - 1;
constexpr int kPackWords = 1024;  // mask words per workgroup in LDS: rpw * ceil(W / 64) <= 1024
constexpr int kPackMaxRows = 16;

// 64-bit add across the lanes of a wavefront (every lane gets the total)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// Wavefront 0 of a pack workgroup: offsets of the mask words inside their rows, the rows' bases (row_base[0..n_here]) and
// the ticket's offset in the packed output from a TWO-LEVEL decoupled look-back.  Tickets form groups of 64:
//   status[t]  = kPackAgg | count of ticket t              (read by the later tickets of the same group)
//   gacc[g]   += (1 << 56) + count, one atomic per ticket: the ticket that completes the group publishes
//   gstat[g]   = kPackAgg | count of group g, later kPackPfx | entries of groups 0..g (by the group's last ticket)
// so a ticket needs ONE load for its own group and one per 64 groups (4096 tickets) behind it, however many workgroups
// are in flight (a flat look-back walked ~28 windows of 64 tickets at 1800 resident workgroups: a third of the kernel).
__device__ __forceinline__ void pack_lookback(int lane, int64_t tk, int64_t n_tickets, int n_here, int n_words,
                                              const unsigned long long* mwords, int* woff, long long* row_base,
                                              unsigned long long* status, unsigned long long* gstat,
                                              unsigned long long* gacc, int64_t* indptr) {
    long long run_rows = 0;
    for (int rr = 0; rr < n_here; ++rr) {
        int run = 0;
        for (int w0 = 0; w0 < n_words; w0 += 64) {
            const int cnt = w0 + lane < n_words ? __popcll(mwords[rr * n_words + w0 + lane]) : 0;
            const int incl = wave_scan_dpp(cnt);
            if (w0 + lane < n_words) woff[rr * n_words + w0 + lane] = run + incl - cnt;
            run += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane == 0) row_base[rr] = run_rows;
        run_rows += run;
    }
    const unsigned long long mine = (unsigned long long)run_rows;
    unsigned long long excl = 0;
#if !(defined(ICV_DEV_EXPERIMENTS) && defined(ICV_PACK_EXP_NOLOOKBACK))  // (experiment: wrong offsets on purpose)
    const int64_t g = tk >> 6;
    const int i = (int)(tk & 63);
    const int in_group = (int)((n_tickets - g * 64) < 64 ? (n_tickets - g * 64) : 64);
    if (lane == 0) {
        __hip_atomic_store(status + tk, kPackAgg | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long old =
            __hip_atomic_fetch_add(gacc + g, (1ull << 56) + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(old >> 56) + 1 == in_group && i != in_group - 1)  // (the group's last ticket publishes the prefix itself)
            __hip_atomic_store(gstat + g, kPackAgg | ((old & ((1ull << 56) - 1)) + mine), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    // own group: the counts of the tickets before this one
    unsigned long long sv = kPackAgg;
    while (true) {
        if (lane < i) sv = __hip_atomic_load(status + g * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64((sv >> 62) == 0) == 0) break;
        __builtin_amdgcn_s_sleep(8);
    }
    excl = wave_sum_u64(lane < i ? (sv & kPackVal) : 0ull);
    // the groups before: 64 per step, up to and including the nearest one whose prefix is known
    int64_t hi = g - 1;
    while (hi >= 0) {
        const int64_t r = hi - lane;
        unsigned long long gv, pfx;
        int first;
        while (true) {
            gv = r >= 0 ? __hip_atomic_load(gstat + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kPackPfx;
            pfx = __builtin_amdgcn_ballot_w64((gv >> 62) == 2);
            first = pfx ? __builtin_ctzll(pfx) : 64;
            if (__builtin_amdgcn_ballot_w64(lane < first && (gv >> 62) == 0) == 0) break;
            __builtin_amdgcn_s_sleep(8);  // pollers cost the streaming workgroups bandwidth: back off
        }
        excl += wave_sum_u64(lane <= first ? (gv & kPackVal) : 0ull);
        if (pfx) break;
        hi -= 64;
    }
    if (lane == 0 && i == in_group - 1)
        __hip_atomic_store(gstat + g, kPackPfx | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (lane == 0) {
        row_base[n_here] = run_rows;
        if (tk == 0) indptr[0] = 0;
    }
    if (lane <= n_here) row_base[lane] += (long long)excl;  // (lane 0's writes above: same wavefront, in order)
}

// (Measured and dropped, profiles/r04_pack_experiments.txt: the 16 rows kept in registers between the decision and the
// packing -- 211 VGPRs, two workgroups per CU; one wavefront per row with 16-byte loads and four ballots per 256
// windows -- 140 VGPRs; non-temporal loads of x_res, which push the packing's second read out of L2.)
template <typename T, bool CSR>
__global__ void __launch_bounds__(256) k_thr_pack(const KParams P, const double* thr, int64_t chunksize,
                                                  int64_t row_phase, int rpw, unsigned int* ticket,
                                                  unsigned long long* status, unsigned long long* gstat,
                                                  unsigned long long* gacc, int64_t* indptr, int32_t* indices,
                                                  double* data, int64_t cap) {
    __shared__ int ticket_s;
    __shared__ unsigned long long mwords[kPackWords], twords[kPackWords];
    __shared__ int woff[kPackWords];
    __shared__ long long row_base[kPackMaxRows + 1];
    if (threadIdx.x == 0) ticket_s = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t tk = ticket_s;
    const int64_t cell0 = tk * rpw;
    const int n_here = (int)((P.n_rows - cell0) < rpw ? (P.n_rows - cell0) : rpw);
    const bool has_thr = thr != nullptr;
    const int n_words = (P.W + 63) >> 6;
    // ---- decide: float32 decides; a value within one ulp of the threshold is kept for now and flagged in a second mask
    // (twords) for the exact float64 recomputation after the pass: the streaming body is a compare and two ballots.
    // The first eight windows per thread of the NEXT row are requested before the current row is decided.
    float nxt[8];
    const auto request = [&](int rr) {
        const float* orow = P.out + (cell0 + (rr < n_here ? rr : 0)) * P.ldo;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = u * 256 + (int)threadIdx.x;
            nxt[u] = (rr < n_here && j < P.W) ? orow[j] : 0.0f;
        }
    };
    request(0);
    for (int rr = 0; rr < n_here; ++rr) {
        const int64_t cell = cell0 + rr;
        const float thf = has_thr ? (float)thr[(cell + row_phase) / chunksize] : 0.0f;
        const float* orow = P.out + cell * P.ldo;
        for (int j0 = 0; j0 < P.W; j0 += 8 * 256) {
            float yv[8];
            if (j0 == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) yv[u] = nxt[u];
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * 256 + (int)threadIdx.x;
                    yv[u] = j < P.W ? orow[j] : 0.0f;
                }
            }
            if (j0 + 8 * 256 >= P.W) request(rr + 1);  // (in flight while this batch is decided)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * 256 + (int)threadIdx.x;
                const float y = yv[u];
                bool keep = y != 0.0f && j < P.W;  // NaN: kept (stored explicitly, like csr_matrix(x_res))
                bool tie = false;
                if (has_thr) {
                    const int cmp = thr_compare(fabsf(y), thf);
                    if (cmp < 0) keep = false;
                    tie = cmp == 0 && keep;
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(keep), tm = __builtin_amdgcn_ballot_w64(tie);
                if ((threadIdx.x & 63) == 0 && j < P.W) {
                    mwords[rr * n_words + (j >> 6)] = m;
                    twords[rr * n_words + (j >> 6)] = tm;
                }
            }
        }
    }
    __syncthreads();  // the mask words of the workgroup's rows are complete
    // ---- ties (about one window in 1e7): each recomputed in float64 by the thread that finds it (one thread per word) ----
    for (int w = threadIdx.x; w < n_here * n_words; w += 256) {
        unsigned long long tm = twords[w];
        while (tm) {
            const int bit = __builtin_ctzll(tm);
            tm &= tm - 1;
            const int rr = w / n_words, j = (w - rr * n_words) * 64 + bit;
            const int64_t cell = cell0 + rr;
            const double th = thr[(cell + row_phase) / chunksize];
            const int st = P.w_start[j];
            const double yd = window_canonical(P, j, [&](int k) { return value_at<T, CSR>(P, cell, st + k); }) -
                              P.cell_median[cell];
            if (fabs(yd) < th) mwords[w] &= ~(1ull << bit);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64)
        pack_lookback(threadIdx.x, tk, (P.n_rows + rpw - 1) / rpw, n_here, n_words, mwords, woff, row_base, status, gstat,
                      gacc, indptr);
    __syncthreads();
    // ---- pack: the rows again (L2), entries in window order ----------------------------------------------------------
    if ((int)threadIdx.x < n_here) indptr[cell0 + threadIdx.x + 1] = row_base[threadIdx.x + 1];
    const int lane = threadIdx.x & 63;
    for (int rr = 0; rr < n_here; ++rr) {
        const float* orow = P.out + (cell0 + rr) * P.ldo;
        const int64_t base = row_base[rr];
        for (int j = threadIdx.x; j < P.W; j += 256) {
            const unsigned long long m = mwords[rr * n_words + (j >> 6)];
            if ((m >> lane) & 1ull) {
                const int64_t pos = base + woff[rr * n_words + (j >> 6)] +
                                    __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_PACK_EXP_NOSTORE)
                if (pos < 0) {
#else
                if (pos < cap) {
#endif
                    indices[pos] = j;
                    data[pos] = (double)orow[j];
                }
            }
        }
    }
}

// x_res = window - median (float32), per-cell moments and median, from float64 windows resident in HBM (the
// last step of the chromosome-group fallback; same DPP reduction tree as the smoothing kernels)
__global__ void __launch_bounds__(256) k_win_finish(const double* win, int64_t n_rows, int W, const double* med,
                                                    float* out, int64_t ldo, double* cell_median,
                                                    double* cell_stats) {
    __shared__ double ps[4], pq[4];
    const int64_t cell = blockIdx.x;
    const double m = med[cell];
    const double* w = win + cell * (int64_t)W;
    float* orow = out + cell * ldo;
    double sum = 0.0, sq = 0.0;
    for (int j = threadIdx.x; j < W; j += 256) {
        const double y = w[j] - m;
        orow[j] = (float)y;
        sum = sum + y;
        sq = fma(y, y, sq);
    }
    sum = wave_sum_dpp(sum);
    sq = wave_sum_dpp(sq);
    if ((threadIdx.x & 63) == 0) {
        ps[threadIdx.x >> 6] = sum;
        pq[threadIdx.x >> 6] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cell_stats[2 * cell] = ((ps[0] + ps[1]) + ps[2]) + ps[3];
        cell_stats[2 * cell + 1] = ((pq[0] + pq[1]) + pq[2]) + pq[3];
        cell_median[cell] = m;
    }
}

// cell_stats[c] = sum over the 8 per-wavefront partial moment pairs of cell c, fixed order (deterministic)
__global__ void __launch_bounds__(256) k_stats_finish(const double* part, int64_t n_rows, double* cell_stats) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_rows) return;
    const double2* p = reinterpret_cast<const double2*>(part) + c * 8;
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const double2 v = p[w];
        s += v.x;
        q += v.y;
    }
    cell_stats[2 * c] = s;
    cell_stats[2 * c + 1] = q;
}

// ---------------------------------------------------------------------------------------
// reference profile: per-group column sums (reference :385, :400), float64
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_colsum_dense(const T* x, int64_t n_rows, int64_t ld, int n_cols,
                                                      const int32_t* row_group, int group, int rows_per_slab,
                                                      double* partial /* n_slabs x n_cols */) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
    int64_t r1 = r0 + rows_per_slab;
    if (r1 > n_rows) r1 = n_rows;
    double acc = 0.0;
    if (col < n_cols) {
        if (row_group == nullptr) {
            int64_t r = r0;
            for (; r + 8 <= r1; r += 8) {
                T v[8];
                // non-temporal: the matrix is streamed once (measured 6.6 against 6.1 TB/s, tools/microbench_colsum.hip)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&x[(r + u) * ld + col]);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (double)v[u];
            }
            for (; r < r1; ++r) acc += (double)x[r * ld + col];
        } else {
            for (int64_t r = r0; r < r1; ++r)
                if (row_group[r] == group) acc += (double)x[r * ld + col];
        }
        partial[(int64_t)blockIdx.y * n_cols + col] = acc;
    }
}

// sums[col] += sum over slabs, in a fixed order: 16 lanes per column each add every 16th slab, then the 16
// partial sums are added in lane order (deterministic; 64 columns x 16 slab lanes per workgroup)
__global__ void __launch_bounds__(1024) k_colsum_finish(const double* partial, int n_slabs, int n_cols, double* sums) {
    __shared__ double part[16][64];
    const int c = threadIdx.x & 63, l = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    double acc = 0.0;
    if (col < n_cols)
        for (int s = l; s < n_slabs; s += 16) acc += partial[(int64_t)s * n_cols + col];
    part[l][c] = acc;
    __syncthreads();
    if (l == 0 && col < n_cols) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += part[q][c];
        sums[col] += tot;
    }
}

// CSR: one 1024-thread workgroup owns (row slab) x (tile of up to 20 000 columns = 160 000 B of float64 accumulators in
// LDS, the whole LDS of a CU).  The rows of the slab are taken one at a time: all sixteen wavefronts add the stored
// entries of the SAME row with ds_add_f64 (the columns of a row are distinct: no two adds meet), then a workgroup
// barrier, then the next row -- every column receives its addends in row order, so the sums are deterministic, unlike
// wavefronts adding different rows into one tile.  The tile then goes to partial[slab][...] and k_colsum_finish adds
// the slabs in a fixed order.
// What makes it fast is the software pipeline: the {column, value} pairs of the next kCsDepth rows are in flight while
// a row is added (a row is ~11 KB, the HBM round trip several rows long), and the row offsets another kCsDepth rows
// ahead of that.  All of these are unconditional loads (buffer loads with the row's length as range; a row that is
// past the slab or belongs to another group has length 0), so the compiler's wait counts are exact.
constexpr int kCsrTileCols = 20000;
constexpr int kCsDepth = 8;
constexpr int kCsThreads = 1024;
template <typename T, bool GROUPS>
__global__ void __launch_bounds__(kCsThreads) k_colsum_csr(const T* vals, const int64_t* indptr, const int32_t* indices,
                                                           int64_t n_rows, int n_cols, const int32_t* row_group, int group,
                                                           int rows_per_slab, double* partial /* n_slabs x n_cols */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    constexpr int D = kCsDepth, NT_ = kCsThreads;
    const int c0 = blockIdx.x * kCsrTileCols;
    const int nc = (n_cols - c0) < kCsrTileCols ? (n_cols - c0) : kCsrTileCols;
    const int t = threadIdx.x;
    for (int i = t; i < nc; i += NT_) tile[i] = 0.0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
    int64_t r1 = r0 + rows_per_slab;
    if (r1 > n_rows) r1 = n_rows;
    if (r0 >= r1) return;  // (uniform)
    int zoff = 0;
    asm volatile("" : "+v"(zoff));  // row offsets through vector loads: scalar loads share their counter with the LDS
    int64_t ipa[D], ipb[D];  // slot s: offsets of the row whose entries are requested next ...
    int ipg[D], ipv[D];      // ... its group, and whether it lies inside the slab
    unsigned ei[D][2];       // slot s: entries in flight (lanes t and t + 1024 of the row) ...
    T ev[D][2];
    int64_t est[D];          // ... the row's first entry and its length (0: nothing to add)
    int cnt[D];
    const auto load_ip = [&](int s, int64_t row) {
        const int64_t rr = row < r1 ? row : r1 - 1;
        ipa[s] = indptr[rr + zoff];
        ipb[s] = indptr[rr + 1 + zoff];
        ipg[s] = GROUPS ? row_group[rr + zoff] : group;
        ipv[s] = row < r1;
    };
    const auto uniform64 = [](int64_t v) {
        return ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    const auto load_entries = [&](int s) {
        const int64_t e0 = uniform64(ipa[s]), e1 = uniform64(ipb[s]);
        const bool mine = ipv[s] && __builtin_amdgcn_readfirstlane(ipg[s]) == group;
        const int64_t n = mine ? e1 - e0 : 0;
        const unsigned npf = (unsigned)(n < 2 * NT_ ? n : 2 * NT_);
        const __amdgpu_buffer_rsrc_t i_rs = make_rsrc(indices + e0, npf * 4u);
        const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vals + e0, npf * (unsigned)sizeof(T));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ei[s][u] = __builtin_amdgcn_raw_buffer_load_b32(i_rs, (unsigned)t * 4u, u * NT_ * 4, 0);
            if constexpr (sizeof(T) == 4) {
                ev[s][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(v_rs, (unsigned)t * 4u, u * NT_ * 4, 0));
            } else {
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(v_rs, (unsigned)t * 8u, u * NT_ * 8, 0);
                ev[s][u] = __hiloint2double((int)w.y, (int)w.x);
            }
        }
        est[s] = e0;
        cnt[s] = (int)(n < 0x7fffffff ? n : 0x7fffffff);
    };
    const auto add = [&](int col, double v) {
        const int c = col - c0;
        if (c >= 0 && c < nc) __hip_atomic_fetch_add(tile + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const auto consume = [&](int s) {
        const int n = cnt[s];
        if (n > 0) {  // (uniform)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (t + u * NT_ < n) add((int)ei[s][u], (double)ev[s][u]);
            for (int k = t + 2 * NT_; k < n; k += NT_) add(indices[est[s] + k], (double)vals[est[s] + k]);  // long rows
            __syncthreads();  // the next row may meet the same columns
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) load_ip(s, r0 + s);
#pragma unroll
    for (int s = 0; s < D; ++s) {
        load_entries(s);
        load_ip(s, r0 + D + s);
    }
    for (int64_t r = r0; r < r1; r += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            consume(s);                  // row r + s
            load_entries(s);             // row r + s + D (its offsets were requested D rows ago)
            load_ip(s, r + s + 2 * D);
        }
    }
    __syncthreads();
    double* dst = partial + (int64_t)blockIdx.y * n_cols + c0;
    for (int i = t; i < nc; i += NT_) dst[i] = tile[i];
}

// cnv_score: per-row sum |x| (tl/_scores.py:66), one wavefront per row, float64
__global__ void __launch_bounds__(256) k_row_abs_sum(const float* x, int64_t n_rows, int n_cols, int64_t ld,
                                                     double* row_sum) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* xr = x + row * ld;
    double acc = 0.0;
    for (int j = threadIdx.x & 63; j < n_cols; j += 64) acc += (double)fabsf(xr[j]);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) row_sum[row] = acc;
}

// ---------------------------------------------------------------------------------------
// calculate_gene_values (reference tl/_infercnv.py:247-298, :443-453)
//   gene value = np.mean over the kept windows that contain the gene (sorted-gene coordinates);
//   then minus the per-cell median over the covered genes; |v| < noise threshold -> 0.
// ---------------------------------------------------------------------------------------
// numpy's float64 add.reduce of a contiguous array (pairwise summation with 8 accumulators for
// 8 <= n <= 128, recursion above) -- np.mean(list of window values) goes through it
__device__ inline double numpy_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;  // numpy starts from -0.0; the sign of a zero sum is irrelevant here
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int i = 8;
        for (; i + 8 <= n; i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_sum(a, n2) + numpy_sum(a + n2, n - n2);
}

// gv[cell][q] for covered gene q: mean of windows j0 .. j0+cnt-1 of that cell
__global__ void __launch_bounds__(256) k_gene_means(const double* win, int64_t n_rows, int W, const int32_t* cov_j0,
                                                    const int32_t* cov_cnt, int n_cov, double* gv) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int64_t cell = blockIdx.y;
    if (q >= n_cov) return;
    const int cnt = cov_cnt[q];
    const double* w = win + cell * (int64_t)W + cov_j0[q];
    gv[cell * (int64_t)n_cov + q] = numpy_sum(w, cnt) / (double)cnt;
}

// per-row median of a float64 matrix resident in HBM (np.median): one workgroup per row, value-space
// bisection with workgroup-wide counts until <= 1024 candidates, which are then ranked in LDS
__global__ void __launch_bounds__(256) k_row_median(const double* x, int64_t n_rows, int n, double* med) {
    __shared__ double cand[1024];
    __shared__ int icnt[2][4];
    __shared__ double dmn[4], dmx[4];
    __shared__ int ncand, anynan;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double* row = x + (int64_t)blockIdx.x * n;
    double mn = __builtin_inf(), mx = -__builtin_inf();
    int nanl = 0;
    for (int i = t; i < n; i += 256) {
        const double v = row[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
        nanl |= (v != v);
    }
    mn = wave_min_dpp(mn);
    mx = wave_max_dpp(mx);
    if (t == 0) { ncand = 0; anynan = 0; }
    __syncthreads();
    if (lane == 0) { dmn[wave] = mn; dmx[wave] = mx; }
    if (nanl) anynan = 1;
    __syncthreads();
    if (anynan) {
        if (t == 0) med[blockIdx.x] = __builtin_nan("");
        return;
    }
    mn = dmn[0]; mx = dmx[0];
    for (int i = 1; i < 4; ++i) { mn = dmn[i] < mn ? dmn[i] : mn; mx = dmx[i] > mx ? dmx[i] : mx; }
    const int k1 = (n - 1) / 2, k2 = n / 2;
    double lo = prev_double(mn), hi = mx;  // order statistics k1, k2 in (lo, hi]
    int cnt_lo = 0, cnt_hi = n, it = 0;
    while (cnt_hi - cnt_lo > 1024 && it < 200) {
        double mid;
        if (it < 60) mid = 0.5 * lo + 0.5 * hi;
        else {
            const unsigned long long a = ordered_key(lo), b = ordered_key(hi);
            mid = from_ordered_key(a + ((b - a) >> 1));
        }
        if (!(mid > lo && mid < hi)) break;
        int c = 0;
        for (int i = t; i < n; i += 256) c += (row[i] <= mid) ? 1 : 0;
        c = wave_sum_i(c);
        if (lane == 0) icnt[it & 1][wave] = c;
        __syncthreads();
        c = icnt[it & 1][0] + icnt[it & 1][1] + icnt[it & 1][2] + icnt[it & 1][3];
        if (c > k2) { hi = mid; cnt_hi = c; }
        else if (c <= k1) { lo = mid; cnt_lo = c; }
        else {  // pivot separates the two middle elements: a = max{<= mid}, b = min{> mid}
            double a = -__builtin_inf(), b = __builtin_inf();
            for (int i = t; i < n; i += 256) {
                const double v = row[i];
                if (v <= mid) a = v > a ? v : a;
                else b = v < b ? v : b;
            }
            a = wave_max_dpp(a);
            b = wave_min_dpp(b);
            __syncthreads();
            if (lane == 0) { dmx[wave] = a; dmn[wave] = b; }
            __syncthreads();
            if (t == 0) {
                for (int i = 1; i < 4; ++i) { a = dmx[i] > a ? dmx[i] : a; b = dmn[i] < b ? dmn[i] : b; }
                med[blockIdx.x] = (a + b) / 2.0;
            }
            return;
        }
        ++it;
    }
    if (cnt_hi - cnt_lo > 1024) {  // adjacent doubles: every candidate equals hi
        if (t == 0) med[blockIdx.x] = hi;
        return;
    }
    for (int i = t; i < n; i += 256) {
        const double v = row[i];
        if (v > lo && v <= hi) {
            const int idx = atomicAdd(&ncand, 1);
            if (idx < 1024) cand[idx] = v;
        }
    }
    __syncthreads();
    const int m = ncand < 1024 ? ncand : 1024;
    // exact ranks: thread i ranks candidates i, i+256, ...
    for (int i = t; i < m; i += 256) {
        const double mine = cand[i];
        int r = 0;
        for (int jj = 0; jj < m; ++jj) {
            const double o = cand[jj];
            r += (o < mine || (o == mine && jj < i)) ? 1 : 0;
        }
        if (r == k1 - cnt_lo) dmn[0] = mine;  // both written before the barrier below, read after
        if (r == k2 - cnt_lo) dmx[0] = mine;
    }
    __syncthreads();
    if (t == 0) med[blockIdx.x] = (k1 == k2) ? dmn[0] : (dmn[0] + dmx[0]) / 2.0;
}

// centre on the per-cell median, zero below the chunk threshold, scatter to input column order
__global__ void __launch_bounds__(256) k_gene_finish(const double* gv, const double* med, const double* thr,
                                                     int64_t chunksize, int64_t row_phase, const int32_t* cov_col,
                                                     int n_cov, double* out, int64_t ldg) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int64_t cell = blockIdx.y;
    if (q >= n_cov) return;
    double v = gv[cell * (int64_t)n_cov + q] - med[cell];
    if (thr) {
        const double th = thr[(cell + row_phase) / chunksize];
        if (fabs(v) < th) v = 0.0;
    }
    out[cell * ldg + cov_col[q]] = v;
}

__global__ void __launch_bounds__(256) k_fill_nan(double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_nan("");
}

// ---------------------------------------------------------------------------------------
// X_cnv is returned as CSR float64 (reference :455, :137): count / pack the non-zeros of the dense
// float32 result on the device, so that only ~13 % of the matrix crosses PCIe.  One wavefront per row.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_csr_count(const float* x, int64_t n_rows, int n_cols, int64_t ld,
                                                   int64_t* row_nnz) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* xr = x + row * ld;
    int c = 0;
    for (int j = threadIdx.x & 63; j < n_cols; j += 64) c += (xr[j] != 0.0f) ? 1 : 0;  // NaN counts as stored
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) row_nnz[row] = c;
}

__global__ void __launch_bounds__(256) k_csr_fill(const float* x, int64_t n_rows, int n_cols, int64_t ld,
                                                  const int64_t* indptr, int32_t* indices, double* data) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * ld;
    int64_t base = indptr[row];
    for (int j0 = 0; j0 < n_cols; j0 += 64) {
        const int j = j0 + lane;
        const float v = j < n_cols ? xr[j] : 0.0f;
        const bool nz = v != 0.0f;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(nz);
        if (nz) {
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            indices[base + pos] = j;
            data[base + pos] = (double)v;
        }
        base += __popcll(m);
    }
}

// cnv_score on a CSR X_cnv: per-row sum |data| over the stored entries (implicit zeros add nothing), float64,
// one wavefront per row, fixed reduction order
template <typename T>
__global__ void __launch_bounds__(256) k_csr_row_abs_sum(const T* data, const int64_t* indptr, int64_t n_rows,
                                                         double* row_sum) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    double acc = 0.0;
    const int64_t e = indptr[row + 1];
    for (int64_t k = indptr[row] + (threadIdx.x & 63); k < e; k += 64) acc += fabs((double)data[k]);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) row_sum[row] = acc;
}

// indptr[0] = 0, indptr[r + 1] = row_nnz[0] + ... + row_nnz[r], in two launches over blocks of kScanBlock rows:
// k_row_block_sums (one workgroup per block: the block's total) and k_row_offsets (one workgroup per block: the totals
// of the blocks before it, then the scan of its own rows).  Round 4 started with ONE workgroup walking all rows (0.1 ms
// per 100 000 rows, latency of 7 dependent sweeps); this form is two ~5 us launches at any row count.
constexpr int kScanPer = 4;
constexpr int kScanBlock = 1024 * kScanPer;
__device__ __forceinline__ long long block_sum_1024(long long s, long long* wsum) {  // every thread gets the total
    s = (long long)wave_sum_u64((unsigned long long)s);
    __syncthreads();  // wsum free again
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    long long tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += wsum[w];
    return tot;
}
__global__ void __launch_bounds__(1024) k_row_block_sums(const int64_t* __restrict__ row_nnz, int64_t n_rows,
                                                         int64_t* __restrict__ block_sums) {
    __shared__ long long wsum[16];
    const int64_t r0 = (int64_t)blockIdx.x * kScanBlock + (int64_t)kScanPer * threadIdx.x;
    long long s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) s += r0 + k < n_rows ? row_nnz[r0 + k] : 0;
    const long long tot = block_sum_1024(s, wsum);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) k_row_offsets(const int64_t* __restrict__ row_nnz, int64_t n_rows,
                                                      const int64_t* __restrict__ block_sums,
                                                      int64_t* __restrict__ indptr) {
    __shared__ long long wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * kScanBlock + (int64_t)kScanPer * t;
    long long v[kScanPer], s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        v[k] = r0 + k < n_rows ? row_nnz[r0 + k] : 0;
        s += v[k];
    }
    long long before = 0;  // the blocks before this one
    for (int b = t; b < (int)blockIdx.x; b += 1024) before += block_sums[b];
    before = block_sum_1024(before, wsum);
    long long incl = s;  // inclusive scan of the threads' sums over the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned lo = (unsigned)__shfl_up((int)(unsigned)incl, o);
        const unsigned hi = (unsigned)__shfl_up((int)(unsigned)((unsigned long long)incl >> 32), o);
        if (lane >= o) incl += (long long)(((unsigned long long)hi << 32) | lo);
    }
    __syncthreads();  // wsum free again
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) before += wsum[w];
    long long run = before + incl - s;
    if (blockIdx.x == 0 && t == 0) indptr[0] = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        run += v[k];
        if (r0 + k < n_rows) indptr[r0 + k + 1] = run;
    }
}

// pack the kept entries (bit mask of k_thr_mask) of the dense float32 result: one wavefront per row.  The row's mask
// words are loaded at once (lane w: word w), their kept counts scanned over the lanes: every word's output offset is
// known before the first value is touched, so the 64-window steps carry no dependency (round 2 walked the words with
// one scalar load and one running offset per step: a memory round trip per 64 windows).
__global__ void __launch_bounds__(256) k_csr_fill_masked(const float* x, int64_t n_rows, int n_cols, int64_t ld,
                                                         const unsigned long long* mask, int n_words,
                                                         const int64_t* indptr, int32_t* indices, double* data) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * ld;
    const unsigned long long* mrow = mask + row * (int64_t)n_words;
    int64_t base = indptr[row];
    for (int w0 = 0; w0 < n_words; w0 += 64) {  // 64 words = 4096 windows per pass
        const int nw = n_words - w0 < 64 ? n_words - w0 : 64;
        const unsigned long long mine = lane < nw ? mrow[w0 + lane] : 0ull;
        const int cnt = __popcll(mine);
        const int incl = wave_scan_dpp(cnt);
        const int excl = incl - cnt;
        const unsigned mlo = (unsigned)mine, mhi = (unsigned)(mine >> 32);
        for (int w = 0; w < nw; ++w) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)mlo, w), hi = (unsigned)__builtin_amdgcn_readlane((int)mhi, w);
            if ((lo | hi) == 0u) continue;  // wavefront-uniform
            const unsigned long long m = ((unsigned long long)hi << 32) | lo;
            const int off = __builtin_amdgcn_readlane(excl, w);
            if ((m >> lane) & 1ull) {
                const int j = (w0 + w) * 64 + lane;
                const int pos = off + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
                indices[base + pos] = j;
                data[base + pos] = (double)xr[j];
            }
        }
        base += __builtin_amdgcn_readlane(incl, 63);
    }
}

}  // namespace icv

// ithcna / ithgex (reference src/infercnvpy/tl/_scores.py:77-221): per group of cells the
// interquartile range of ALL entries of the cell x cell Pearson correlation matrix
//   pcorr = np.corrcoef(X)            (:138, :207)
//   q75, q25 = np.percentile(pcorr, [75, 25]);  score = q75 - q25     (:143-144, :212-213)
// This is the one dense contraction next to the hot path: rows are centred and scaled to unit
// length (float64 statistics, float32 result), the Gram matrix Z Z^T is computed with fp32 MFMA
// tiles (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains), and the four order statistics the two
// percentiles interpolate between are selected exactly by bisection on the ordered float32 keys.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "icv_kernels.hpp"

namespace icv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// z[i] = (x[i] - mean_i) / ||x[i] - mean_i||  (one wavefront per row; constant rows -> NaN as numpy 0/0)
__global__ void __launch_bounds__(256) k_row_normalize(const float* x, int64_t n, int k, int64_t ld, float* z,
                                                       int kz /* padded row length of z, zero filled */) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * ld;
    double s = 0.0;
    for (int j = lane; j < k; j += 64) s += (double)xr[j];
    s = wave_sum_dpp(s);
    const double mean = s / (double)k;
    double q = 0.0;
    for (int j = lane; j < k; j += 64) {
        const double d = (double)xr[j] - mean;
        q = fma(d, d, q);
    }
    q = wave_sum_dpp(q);
    const double inv = 1.0 / sqrt(q);  // q == 0 -> inf -> 0 * inf = NaN
    float* zr = z + row * (int64_t)kz;
    for (int j = lane; j < kz; j += 64) zr[j] = j < k ? (float)(((double)xr[j] - mean) * inv) : 0.0f;
}

// C = Z Z^T, Z row-major n x kz (kz multiple of 16, zero padded), C row-major float32.
// Workgroup = 256 threads = 4 wavefronts, 128 x 128 output tile, each wavefront a 64 x 64 quadrant as
// 2 x 2 MFMA tiles of 32 x 32; K advances 16 per LDS stage.  mfma_f32_32x32x2f32 operand layout:
// A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; D: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Operand panels in LDS: row stride 20 floats (16-byte aligned rows; a 16-lane group of a ds_read_b128 covers
// the 64 banks exactly once).  A lane reads four consecutive K values of its row with one ds_read_b128 and
// feeds them to four MFMAs: the lower half of the wavefront takes K = 8g .. 8g+3, the upper half 8g+4 .. 8g+7
// (a permutation of the summation index, the same for both operands).  The panels are double buffered: the
// global loads of stage s+2 are in flight (registers) and stage s+1 is written to the other buffer while stage s
// is multiplied -- one barrier per stage.
//
// Instruction order is pinned (sched_barrier): the MFMA pipe issues in order and the two wavefronts that share a
// SIMD run in lock step (same code, same barriers), so a wait that is exposed in one wavefront is exposed in
// both.  Every LDS read is issued >= 8 MFMAs (512 cycles) before its first use, the LDS writes 8 MFMAs before
// the barrier, the global loads a whole stage before they are written to LDS.  Measured (tools/bench_gram.hip):
// the MFMA sequence alone sustains 155 TFLOP/s; with compiler-placed waits the same kernel reached 112.
//
// Accumulation is two-level: every 32 x 32 tile's MFMA chain runs over 128 columns of K (8 stages) and is then
// added to a second float32 accumulator -- the rounding walk of a K = 5000 chain is ~6x shorter.  The four
// tiles of a wavefront are flushed in different stages (tile T in stage 2T of every block of 8): its 16 adds sit
// in the shadow of the other tiles' MFMAs and the chain restarts from a literal zero C operand.  Both accumulator
// sets live in VGPRs (-mllvm -amdgpu-mfma-vgpr-form: no v_accvgpr traffic; the first version of this kernel moved
// all 64 accumulators between the register classes in every stage).
// DIST = false: C = clip(Z Z^T, -1, 1) (np.corrcoef); DIST = true: C[i][j] = max(0, |z_i|^2 + |z_j|^2 - 2 z_i.z_j),
// 0 on the diagonal (squared Euclidean distances; norm[] float64 from k_center_rows).
//
// Work list: super-tiles of 8 x 8 tiles (1024 x 1024 outputs), GramSuper descriptors built by the host.  The grid
// is one-dimensional; workgroup id -> XCD = id % 8 (round-robin dispatch), and XCD x takes the super-tiles
// x, x + 8, ...: the 64 tiles in flight on an XCD share 8 + 8 operand panels through its L2.
// SYM = true: tiles below the diagonal are skipped; an off-diagonal tile is also written transposed (through a
// 32 x 33 LDS patch per wavefront, so both writes are coalesced) to the mirror target -- half the flops and a
// bit-exactly symmetric matrix.  Direct and mirror targets are separate (pointer, stride, offset per super-tile):
// one GPU writes both into the same n x n matrix, a rank of a sharded job writes its own rows directly and the
// mirror blocks into the buffer that travels to the owners of those rows (dist.py).
// The value of an entry depends only on the unordered pair {i, j} (same K order, flush stages symmetric in the
// tile coordinates), so separately computed mirror images agree bit for bit as well.
//
// Measured on MI355X (tools/bench_gram.hip, n = 32768, K = 5008): 142.8 TFLOP/s executed = 0.91 of the fp32 MFMA
// peak (round 1: 0.69; compiler-placed waits 0.71; pinned order on the old two-dimensional grid with its empty
// below-diagonal workgroups 0.80).  A start-up skew of the two workgroups sharing a CU (so that their epilogues
// do not coincide) was measured on top of this and changed nothing.
struct GramSuper {
    int row0, col0;   // first row / column of the super-tile
    int64_t dir_off;  // element offset of (row0, col0) in the direct target
    int64_t mir_off;  // element offset of the transposed block's origin in the mirror target
};
struct GramJob {
    const GramSuper* supers;
    int n_supers;
    int64_t row_end, col_end;  // valid rows / columns
    float* c_dir;
    int64_t ld_dir;
    float* c_mir;
    int64_t ld_mir;
};
constexpr int GT = 128, GK = 16, GLD = 20, GSUPER = 8;
constexpr int GPANEL = GT * GLD;  // floats per operand panel
template <int V>
using gint = std::integral_constant<int, V>;

#define ICV_SB() __builtin_amdgcn_sched_barrier(0)

template <bool DIST, bool SYM, int EXP = 0>  // EXP != 0: tools/bench_gram.hip experiments (wrong results)
__global__ void __launch_bounds__(256) k_gram_mfma(const float* z, int kz, const double* norm, const GramJob J) {
    __shared__ __attribute__((aligned(16))) float smem[4 * GPANEL];  // {A, B} x 2 buffers: 40 KB
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seq = blockIdx.x >> 3;
    const int sidx = (seq >> 6) * 8 + (blockIdx.x & 7);
    if (sidx >= J.n_supers) return;
    const GramSuper S = J.supers[sidx];
    const int qy = (seq >> 3) & 7, qx = seq & 7;
    const int64_t i0 = (int64_t)S.row0 + qy * GT, j0 = (int64_t)S.col0 + qx * GT;
    if (i0 >= J.row_end || j0 >= J.col_end || (SYM && j0 < i0)) return;
    const int64_t row1 = J.row_end, n = J.col_end;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
    f32x16 acc[4], tot[4];  // tile T = 2 a + b: rows wi + 32 a, columns wj + 32 b
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        acc[T] = zero;
        tot[T] = zero;
    }

    // this thread stages rows r = t / 4 and r + 64, columns k0 + 4 (t % 4) .. + 3 of both panels.  Buffer loads
    // against one descriptor per panel (its valid rows only: rows past the end read zeros without traffic); the
    // per-thread offsets are two registers, the stage offset is scalar.
    const int64_t rows_a = row1 - i0 < GT ? row1 - i0 : GT, rows_b = n - j0 < GT ? n - j0 : GT;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(z + i0 * (int64_t)kz, (unsigned)(rows_a * kz * 4));
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(z + j0 * (int64_t)kz, (unsigned)(rows_b * kz * 4));
    // (the scalar offset of a buffer load is not range checked: the row half goes into the vector offset)
    unsigned g_off[2];
    g_off[0] = (unsigned)(((t >> 2) * kz + (t & 3) * 4) * 4);
    g_off[1] = g_off[0] + (unsigned)(64 * kz * 4);
    u32x4 va[2], vb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            va[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, g_off[h], (unsigned)k0 * 4u, 0);
            vb[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, g_off[h], (unsigned)k0 * 4u, 0);
        }
    };
    const int st_off = (t >> 2) * GLD + (t & 3) * 4;
    auto store = [&](int buf) {
        float* sa = smem + buf * 2 * GPANEL + st_off;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<u32x4*>(sa + h * 64 * GLD) = va[h];
            *reinterpret_cast<u32x4*>(sa + GPANEL + h * 64 * GLD) = vb[h];
        }
    };
    const int a_off = (wi + (lane & 31)) * GLD + 4 * (lane >> 5);
    const int b_off = GPANEL + (wj + (lane & 31)) * GLD + 4 * (lane >> 5);
    float4 av[2][2], bv[2][2];  // [slot][tile row / tile column]
    auto operands = [&](int buf, int g, int slot) {
        const float* s = smem + buf * 2 * GPANEL + 8 * g;
#pragma unroll
        for (int a = 0; a < 2; ++a) av[slot][a] = *reinterpret_cast<const float4*>(s + a_off + a * 32 * GLD);
#pragma unroll
        for (int b = 0; b < 2; ++b) bv[slot][b] = *reinterpret_cast<const float4*>(s + b_off + b * 32 * GLD);
    };
    auto comp = [](const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
    // one MFMA: tile T, K element e of operand slot `slot`; fresh: the chain restarts (C = 0)
    auto mf = [&](int slot, int T, int e, bool fresh) {
        const float x = comp(av[slot][T >> 1], e), y = comp(bv[slot][T & 1], e);
        if (fresh)
            acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else
            acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[T], 0, 0, 0);
    };

    // One stage (16 columns of K) from buffer cur.  Operand slot 0 holds group 0 of this stage on entry and
    // group 0 of the next stage on exit.  FM: tiles flushed to the second-level sums in this stage (bit mask);
    // ZM: tiles whose chains restart from zero in this stage (flushed in the previous one).
    auto stage = [&](int k0, int cur, bool more, bool more2, auto FM_, auto ZM_) {
        constexpr int FM = decltype(FM_)::value, ZM = decltype(ZM_)::value;
        constexpr int NF = ((FM >> 0) & 1) + ((FM >> 1) & 1) + ((FM >> 2) & 1) + ((FM >> 3) & 1);
#pragma unroll
        for (int T = 0; T < 4; ++T) mf(0, T, 0, (ZM >> T) & 1);
        ICV_SB();
        if (EXP < 2) operands(cur, 1, 1);
        ICV_SB();
#pragma unroll
        for (int T = 0; T < 4; ++T) mf(0, T, 1, false);
        ICV_SB();
        if (more) {
            if (EXP < 2) store(cur ^ 1);  // the buffer stage s-1 was read from (all its reads precede the last barrier)
            if (EXP == 0 && more2) fetch(k0 + 2 * GK);
        }
        ICV_SB();
#pragma unroll
        for (int e = 2; e < 4; ++e)
#pragma unroll
            for (int T = 0; T < 4; ++T) mf(0, T, e, false);
        ICV_SB();
        if (EXP < 3) __syncthreads();
        ICV_SB();
        // group 1: the flushed tiles' MFMAs first; their adds follow >= 4 MFMAs later in the shadow of the others'
        auto rest = [&](int e) {
#pragma unroll
            for (int T = 0; T < 4; ++T)
                if (!((FM >> T) & 1)) mf(1, T, e, false);
        };
        auto flush = [&](int half) {
#pragma unroll
            for (int T = 0; T < 4; ++T)
                if ((FM >> T) & 1) {
#pragma unroll
                    for (int r = 8 * half; r < 8 * half + 8; ++r) tot[T][r] += acc[T][r];
                }
        };
        if constexpr (NF > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int T = 0; T < 4; ++T)
                    if ((FM >> T) & 1) mf(1, T, e, false);
        }
        rest(0);
        ICV_SB();
        if (more && EXP < 2) operands(cur ^ 1, 0, 0);
        ICV_SB();
        rest(1);
        ICV_SB();
        if constexpr (NF > 0) flush(0);
        ICV_SB();
        rest(2);
        ICV_SB();
        if constexpr (NF > 0) flush(1);
        ICV_SB();
        rest(3);
        ICV_SB();
    };

    const int ns = kz / GK;
    fetch(0);
    store(0);
    if (ns > 1) fetch(GK);
    __syncthreads();
    operands(0, 0, 0);
    int s = 0;
    for (; s + 10 <= ns; s += 8) {  // 8 stages, every tile flushed once; stages s+8, s+9 exist: no bounds tests
        const int k0 = s * GK;
        // tile T = 2 a + b; tiles 1 and 2 (mirror images of each other) are flushed in the same stage
        stage(k0, 0, true, true, gint<1>{}, gint<0>{});
        stage(k0 + GK, 1, true, true, gint<0>{}, gint<1>{});
        stage(k0 + 2 * GK, 0, true, true, gint<0>{}, gint<0>{});
        stage(k0 + 3 * GK, 1, true, true, gint<6>{}, gint<0>{});
        stage(k0 + 4 * GK, 0, true, true, gint<0>{}, gint<6>{});
        stage(k0 + 5 * GK, 1, true, true, gint<0>{}, gint<0>{});
        stage(k0 + 6 * GK, 0, true, true, gint<8>{}, gint<0>{});
        stage(k0 + 7 * GK, 1, true, true, gint<0>{}, gint<8>{});
    }
    for (; s < ns; ++s) stage(s * GK, s & 1, s + 1 < ns, s + 2 < ns, gint<0>{}, gint<0>{});
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[T][r] += acc[T][r];
    const bool mirror = SYM && i0 != j0;
    __syncthreads();  // the last stage has no trailing barrier of its own
    float* patch = smem + wave * (32 * 33);  // 4 x 4224 B <= the operand panels (dead now)
    float* cd = J.c_dir + S.dir_off;
    float* cm = J.c_mir + S.mir_off;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t row = i0 + wi + a * 32 + rl;
                const int64_t col = j0 + wj + b * 32 + (lane & 31);
                float v = tot[2 * a + b][r];
                if (row < row1 && col < n) {
                    if (DIST) {
                        const double d2 = norm[row] + norm[col] - 2.0 * (double)v;
                        v = (row == col || !(d2 > 0.0)) ? (d2 != d2 ? (float)d2 : 0.0f) : (float)d2;
                    } else {
                        v = v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v);  // np.corrcoef clips to [-1, 1]
                    }
                    cd[(row - S.row0) * J.ld_dir + (col - S.col0)] = v;
                }
                if (mirror) patch[rl * 33 + (lane & 31)] = v;
            }
            if (mirror) {
                // transposed copy: output row = column cl of the sub-tile, 32 contiguous entries (two rows per pass)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int cl = 2 * q + (lane >> 5);
                    const int64_t orow = j0 + wj + b * 32 + cl;           // a column of the tile above the diagonal
                    const int64_t ocol = i0 + wi + a * 32 + (lane & 31);  // its row index
                    const float v = patch[(lane & 31) * 33 + cl];
                    if (orow < n && ocol < row1) cm[(orow - S.col0) * J.ld_mir + (ocol - S.row0)] = v;
                }
            }
        }
}

__device__ __forceinline__ unsigned ordered_key32(float x) {
    const unsigned b = __float_as_uint(x);
    return (b >> 31) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float from_ordered_key32(unsigned k) {
    const unsigned b = (k >> 31) ? (k & 0x7fffffffu) : ~k;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
#endif
}

// Radix selection of order statistics over the ordered 32-bit keys: one pass histograms the `nbits` key bits at `shift` of
// the elements whose higher bits equal one of up to four prefixes (himask = the bits above the digit; nq histograms of
// 2048 64-bit counters in `hist`, per-workgroup counts collected in LDS first); hist[nq * 2048] counts the NaNs.  Three
// passes (11 + 11 + 10 bits) locate four ranks exactly where the bisection on k_count_le4 needed 33 passes over the matrix.
__global__ void __launch_bounds__(256) k_key_hist(const float* __restrict__ v, int64_t m, int shift, int nbits,
                                                  unsigned himask, unsigned p0, unsigned p1, unsigned p2, unsigned p3,
                                                  int nq, unsigned long long* __restrict__ hist) {
    __shared__ unsigned lh[4 * 2048 + 1];
    for (int i = threadIdx.x; i < 4 * 2048 + 1; i += 256) lh[i] = 0u;
    __syncthreads();
    const unsigned dmask = (1u << nbits) - 1u;
    const unsigned pq[4] = {p0, p1, p2, p3};
    const auto one = [&](float x) {
        if (x != x) {
            atomicAdd(&lh[4 * 2048], 1u);
            return;
        }
        const unsigned k = ordered_key32(x);
        const unsigned d = (k >> shift) & dmask;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq && ((k ^ pq[q]) & himask) == 0u) atomicAdd(&lh[q * 2048 + d], 1u);
    };
    // 16-byte loads, four of them in flight per thread (one 4-byte load per thread and trip left a CU with 4 KB in
    // flight: 1.46 ms per pass over the 2.5 GB of a 25 000-cell group = 1.7 TB/s)
    const bool vec = (reinterpret_cast<uintptr_t>(v) & 15) == 0;
    const int64_t m4 = vec ? m / 4 : 0;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t* v4 = reinterpret_cast<const f32x4_t*>(v);
    const int64_t stride = (int64_t)gridDim.x * 256;
    constexpr int U = 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m4; i += U * stride) {
        f32x4_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < m4) x[u] = __builtin_nontemporal_load(v4 + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < m4) {
                one(x[u].x);
                one(x[u].y);
                one(x[u].z);
                one(x[u].w);
            }
    }
    for (int64_t i = 4 * m4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) one(v[i]);
    __syncthreads();
    for (int i = threadIdx.x; i < nq * 2048; i += 256)
        if (lh[i]) atomicAdd(hist + i, (unsigned long long)lh[i]);
    if (threadIdx.x == 0 && lh[4 * 2048]) atomicAdd(hist + 4 * 2048, (unsigned long long)lh[4 * 2048]);
}

// counts[p] += #{ key(v) <= pivot[p] }, p < 4; counts[4] += #NaN
__global__ void __launch_bounds__(256) k_count_le4(const float* v, int64_t m, unsigned p0, unsigned p1, unsigned p2,
                                                   unsigned p3, unsigned long long* counts) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, cn = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const float x = v[i];
        if (x != x) { ++cn; continue; }
        const unsigned k = ordered_key32(x);
        c0 += k <= p0;
        c1 += k <= p1;
        c2 += k <= p2;
        c3 += k <= p3;
    }
    __shared__ unsigned long long sh[5][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto wsum = [](unsigned long long x) {
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        return x;
    };
    c0 = wsum(c0); c1 = wsum(c1); c2 = wsum(c2); c3 = wsum(c3); cn = wsum(cn);
    if (lane == 0) { sh[0][wave] = c0; sh[1][wave] = c1; sh[2][wave] = c2; sh[3][wave] = c3; sh[4][wave] = cn; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const unsigned long long s = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
        if (s) atomicAdd(counts + threadIdx.x, s);
    }
}

}  // namespace icv

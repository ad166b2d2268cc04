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

// C = Z Z^T, Z row-major n x kz (kz multiple of 16, zero padded), C row-major n x n float32.
// Workgroup = 256 threads = 4 wavefronts, 128 x 128 output tile, each wavefront a 64 x 64 quadrant as
// 2 x 2 MFMA tiles of 32 x 32; K advances 16 per LDS stage.  mfma_f32_32x32x2f32 operand layout:
// A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; D: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// DIST = false: C = clip(Z Z^T, -1, 1) (np.corrcoef); DIST = true: C[i][j] = max(0, |z_i|^2 + |z_j|^2 - 2 z_i.z_j),
// 0 on the diagonal (squared Euclidean distances; norm[] float64 from k_center_rows).
constexpr int GT = 128, GK = 16, GLD = GK + 1;  // +1: conflict-free column reads
// Rows [row0, row0 + gridDim.y * 128) of the result are produced (row block of a sharded matrix); c points at
// the first of them.  Accumulation is two-level: the MFMA chain runs over GSEG columns of K, then is added to
// a second float32 accumulator -- the rounding walk of a K = 5000 chain is ~6x shorter.
constexpr int GSEG = 128;
template <bool DIST>
__global__ void __launch_bounds__(256) k_gram_mfma(const float* z, int64_t n, int kz, float* c, int64_t ldc,
                                                   const double* norm, int64_t row0, int64_t row1) {
    __shared__ float sa[GT * GLD], sb[GT * GLD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i0 = row0 + (int64_t)blockIdx.y * GT, j0 = (int64_t)blockIdx.x * GT;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[a][b][r] = 0.0f;
                tot[a][b][r] = 0.0f;
            }

    for (int k0 = 0; k0 < kz; k0 += GK) {
        // stage 128 rows x 16 columns of both panels: 2048 floats each, 8 per thread (two float4)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = t + h * 256;       // 0..511: row = idx / 4, quarter = idx % 4
            const int r = idx >> 2, qd = idx & 3;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (i0 + r < row1) va = *reinterpret_cast<const float4*>(z + (i0 + r) * (int64_t)kz + k0 + qd * 4);
            if (j0 + r < n) vb = *reinterpret_cast<const float4*>(z + (j0 + r) * (int64_t)kz + k0 + qd * 4);
            float* pa = sa + r * GLD + qd * 4;
            float* pb = sb + r * GLD + qd * 4;
            pa[0] = va.x; pa[1] = va.y; pa[2] = va.z; pa[3] = va.w;
            pb[0] = vb.x; pb[1] = vb.y; pb[2] = vb.z; pb[3] = vb.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const int kc = kk + (lane >> 5);
            float av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = sa[(wi + a * 32 + (lane & 31)) * GLD + kc];
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[b] = sb[(wj + b * 32 + (lane & 31)) * GLD + kc];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (((k0 + GK) & (GSEG - 1)) == 0 || k0 + GK >= kz) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        tot[a][b][r] += acc[a][b][r];
                        acc[a][b][r] = 0.0f;
                    }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = i0 + wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t col = j0 + wj + b * 32 + (lane & 31);
                if (row < row1 && col < n) {
                    float v = tot[a][b][r];
                    if (DIST) {
                        const double d2 = norm[row] + norm[col] - 2.0 * (double)v;
                        v = (row == col || !(d2 > 0.0)) ? (d2 != d2 ? (float)d2 : 0.0f) : (float)d2;
                    } else {
                        v = v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v);  // np.corrcoef clips to [-1, 1]
                    }
                    c[(row - row0) * ldc + col] = v;
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

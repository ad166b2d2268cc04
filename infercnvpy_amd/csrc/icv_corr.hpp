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

// C = Z Z^T, Z row-major n x kz (kz multiple of 16, zero padded), C row-major float32.
// Workgroup = 256 threads = 4 wavefronts, 128 x 128 output tile, each wavefront a 64 x 64 quadrant as
// 2 x 2 MFMA tiles of 32 x 32; K advances 16 per LDS stage, the next stage's global loads are in flight
// (registers) while the current one is multiplied.  mfma_f32_32x32x2f32 operand layout:
// A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; D: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// DIST = false: C = clip(Z Z^T, -1, 1) (np.corrcoef); DIST = true: C[i][j] = max(0, |z_i|^2 + |z_j|^2 - 2 z_i.z_j),
// 0 on the diagonal (squared Euclidean distances; norm[] float64 from k_center_rows).
// SYM = true (square result, row0 = 0): only tiles on or above the diagonal are computed; an off-diagonal tile
// is also written transposed (through a 32 x 33 LDS patch per wavefront, so both writes are coalesced) --
// half the flops and a bit-exactly symmetric matrix.  SYM = false: rows [row0, row1) against all columns (the
// row block of a sharded matrix; c points at row row0).
// Accumulation is two-level: the MFMA chain runs over GSEG columns of K, then is added to a second float32
// accumulator -- the rounding walk of a K = 5000 chain is ~6x shorter.
constexpr int GT = 128, GK = 16, GLD = GK + 1;  // +1: conflict-free column reads
constexpr int GSEG = 128;
template <bool DIST, bool SYM>
__global__ void __launch_bounds__(256) k_gram_mfma(const float* z, int64_t n, int kz, float* c, int64_t ldc,
                                                   const double* norm, int64_t row0, int64_t row1) {
    __shared__ float smem[2 * GT * GLD];
    if (SYM && blockIdx.x < blockIdx.y) return;  // the mirror image of tile (x, y) is written by tile (y, x)
    float* sa = smem;
    float* sb = smem + GT * GLD;
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

    // this thread stages rows r = idx / 4 (idx = t, t + 256), columns k0 + 4 (idx % 4) .. + 3 of both panels
    float4 va[2], vb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = t + h * 256;
            const int r = idx >> 2, qd = idx & 3;
            va[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            vb[h] = va[h];
            if (i0 + r < row1) va[h] = *reinterpret_cast<const float4*>(z + (i0 + r) * (int64_t)kz + k0 + qd * 4);
            if (j0 + r < n) vb[h] = *reinterpret_cast<const float4*>(z + (j0 + r) * (int64_t)kz + k0 + qd * 4);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kz; k0 += GK) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = t + h * 256;
            const int r = idx >> 2, qd = idx & 3;
            float* pa = sa + r * GLD + qd * 4;
            float* pb = sb + r * GLD + qd * 4;
            pa[0] = va[h].x; pa[1] = va[h].y; pa[2] = va[h].z; pa[3] = va[h].w;
            pb[0] = vb[h].x; pb[1] = vb[h].y; pb[2] = vb[h].z; pb[3] = vb[h].w;
        }
        __syncthreads();
        if (k0 + GK < kz) fetch(k0 + GK);
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
    const bool mirror = SYM && blockIdx.x != blockIdx.y;
    float* patch = smem + wave * (32 * 33);  // 4 x 4224 B <= the operand panels (dead after the last barrier)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t row = i0 + wi + a * 32 + rl;
                const int64_t col = j0 + wj + b * 32 + (lane & 31);
                float v = tot[a][b][r];
                if (row < row1 && col < n) {
                    if (DIST) {
                        const double d2 = norm[row] + norm[col] - 2.0 * (double)v;
                        v = (row == col || !(d2 > 0.0)) ? (d2 != d2 ? (float)d2 : 0.0f) : (float)d2;
                    } else {
                        v = v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v);  // np.corrcoef clips to [-1, 1]
                    }
                    c[(row - row0) * ldc + col] = v;
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
                    if (orow < n && ocol < row1) c[orow * ldc + ocol] = v;
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

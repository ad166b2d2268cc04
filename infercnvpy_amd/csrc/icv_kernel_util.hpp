// Small kernels around the device-resident CSR X_cnv (`infercnvpy_amd.PackedCsr`) and user-built device CSR input:
//   k_csr_check      a device CSR matrix is what the kernels expect (offsets monotone and inside the buffers, column
//                    indices in range, ascending and unique within a row) -- the host path checks the same through
//                    scipy's has_canonical_format (reference tl/_infercnv.py:115-116 converts with tocsr())
//   k_csr_densify    selected CSR rows -> dense float32 tile (the input of k_gram_mfma: cell x cell correlations of
//                    tl.ithcna, reference tl/_scores.py:197-213, and the config-5 distances) without leaving HBM
//   k_group_sums     per-group sums of per-cell values in a FIXED order (cnv_score: mean |x| per group, reference
//                    tl/_scores.py:65-68); one workgroup per group, no atomics: host-input and device-input calls give
//                    the same bits
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace icv {

enum { kCsrBadOffsets = 1, kCsrBadColumn = 2, kCsrUnsorted = 4 };

// one wavefront per row; flag |= the defects found (capacity = entries the index / value buffers hold)
__global__ void __launch_bounds__(256) k_csr_check(const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices, int64_t n_rows, int n_cols,
                                                   int64_t capacity, int* __restrict__ flag) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t e0 = indptr[row], e1 = indptr[row + 1];
    int bad = 0;
    if (e0 < 0 || e1 < e0 || e1 > capacity) {
        bad = kCsrBadOffsets;  // (nothing of this row is dereferenced)
    } else {
        for (int64_t k = e0 + lane; k < e1; k += 64) {
            const int c = indices[k];
            if (c < 0 || c >= n_cols) bad |= kCsrBadColumn;
            if (k > e0 && indices[k - 1] >= c) bad |= kCsrUnsorted;
        }
    }
    if (bad) atomicOr(flag, bad);
}

// out[q][c] = value of (rows[q], c), zeros elsewhere; `out` was zeroed by the caller (hipMemsetAsync); one wavefront per
// selected row, entries scattered (columns of a row are distinct)
template <typename T>
__global__ void __launch_bounds__(256) k_csr_densify(const T* __restrict__ data, const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices,
                                                     const int64_t* __restrict__ rows, int64_t n_sel,
                                                     float* __restrict__ out, int64_t ldo) {
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n_sel) return;
    const int64_t row = rows ? rows[q] : q;
    float* o = out + q * ldo;
    const int64_t e1 = indptr[row + 1];
    for (int64_t k = indptr[row] + (threadIdx.x & 63); k < e1; k += 64) o[indices[k]] = (float)data[k];
}

// out[r][c] = value of the stored entry (r, c) in the matrix dtype; `out` was zeroed by the caller.  The device half of
// the sparse upload of a mostly-zero dense host matrix: the dense slab the smoothing kernels read is rebuilt in HBM from
// 8 bytes per stored entry instead of crossing PCIe as 4 bytes per element.
template <typename T>
__global__ void __launch_bounds__(256) k_csr_scatter_rows(const T* __restrict__ data, const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices, int64_t n_rows,
                                                          T* __restrict__ out, int64_t ldo) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    T* o = out + row * ldo;
    const int64_t e1 = indptr[row + 1];
    for (int64_t k = indptr[row] + (threadIdx.x & 63); k < e1; k += 64) o[indices[k]] = data[k];
}

// sums[g] = sum of values[i] over i with group[i] == g, counts[g] = how many: thread t adds its rows t, t + 1024, ... in
// order, then a fixed tree over the 1024 partial sums (the result depends on nothing but the inputs)
__global__ void __launch_bounds__(1024) k_group_sums(const double* __restrict__ values,
                                                     const int32_t* __restrict__ group, int64_t n,
                                                     double* __restrict__ sums, int64_t* __restrict__ counts) {
    __shared__ double s_sum[1024];
    __shared__ long long s_cnt[1024];
    const int g = blockIdx.x;
    double acc = 0.0;
    long long cnt = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if (group[i] == g) {
            acc += values[i];
            ++cnt;
        }
    s_sum[threadIdx.x] = acc;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[g] = s_sum[0];
        counts[g] = s_cnt[0];
    }
}

}  // namespace icv

// BASELINE config 5: cell x cell squared Euclidean distances of X_cnv
// (fp32 MFMA tiles, icv_corr.hpp) and Ward linkage on the resident distance matrix.
//
// The reference has no call site for this (scanpy's heatmap dendrogram works on category means); the
// oracle is scipy: pdist(X) + linkage(y, "ward").  Ward is a reducible linkage, so every pair of
// reciprocal nearest neighbours (RNN) is a merge of the final dendrogram and all RNN pairs of one
// round can be merged at once -- the whole matrix is processed by data-parallel passes instead of
// scipy's sequential nearest-neighbour chain:
//
//   round:  k_ward_round   one workgroup per live row: apply the previous round's merges to the row
//                          (Lance-Williams in float64 on the float32 squared distances, in place),
//                          and find the row's nearest live neighbour
//           k_ward_pairs   one workgroup: detect RNN pairs in row order (deterministic), append them to
//                          the merge log, set the column roles for the next round, compact the live list
//
// Distances are stored squared: Ward's update is linear in d^2
//   d2(k, i+j) = ((n_i+n_k) d2(k,i) + (n_j+n_k) d2(k,j) - n_k d2(i,j)) / (n_i+n_j+n_k)
// and heights are sqrt(d2) at the end (scipy's convention: d(A,B)^2 = 2|A||B|/(|A|+|B|) ||c_A - c_B||^2).
// Entries between two clusters that both merged in the same round are evaluated through one canonical
// order (lower slot first), so the matrix stays bit-exactly symmetric without transposed writes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "icv_kernels.hpp"

namespace icv {

// ---- centring prologue of the distance matrix ----------------------------------------------------
// partial[s][j] = sum of x[i][j] over row slab s (float64), slabs in fixed order -> deterministic mean
__global__ void __launch_bounds__(256) k_colsum_slabs(const float* x, int64_t n, int d, int64_t ld, int n_slabs,
                                                      double* partial) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const int64_t per = (n + n_slabs - 1) / n_slabs;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < n ? r0 + per : n;
    double s = 0.0;
    for (int64_t i = r0; i < r1; ++i) s += (double)x[i * ld + j];
    partial[(int64_t)blockIdx.y * d + j] = s;
}
__global__ void __launch_bounds__(256) k_colmean_finish(const double* partial, int64_t n, int d, int n_slabs,
                                                        double* mean) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    double s = 0.0;
    for (int b = 0; b < n_slabs; ++b) s += partial[(int64_t)b * d + j];
    mean[j] = s / (double)n;
}
// z[i] = float(x[i] - mean) padded with zeros to kz; norm[i] = sum z^2 (float64, of the ROUNDED z so that
// d2(i,i) cancels exactly against the MFMA dot product's leading terms)
__global__ void __launch_bounds__(256) k_center_rows(const float* x, int64_t n, int d, int64_t ld, const double* mean,
                                                     float* z, int kz, double* norm) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * ld;
    float* zr = z + row * (int64_t)kz;
    double q = 0.0;
    for (int j = lane; j < kz; j += 64) {
        float v = 0.0f;
        if (j < d) v = (float)((double)xr[j] - mean[j]);
        zr[j] = v;
        q = fma((double)v, (double)v, q);
    }
    q = wave_sum_dpp(q);
    if (lane == 0) norm[row] = q;
}

// ---- Ward rounds ------------------------------------------------------------------------------------
struct WardCounts {
    int n_live, n_merges, n_pairs, pad;
};

__device__ __forceinline__ float ward_lw(float dac, float dbc, float dab, int na, int nb, int nc) {
    const double t = (double)(na + nb + nc);
    const double v = ((double)(na + nc) * (double)dac + (double)(nb + nc) * (double)dbc - (double)nc * (double)dab) / t;
    return v > 0.0 ? (float)v : 0.0f;
}

// One workgroup per live row r (slot index): bring the row up to date with the merges of the previous
// round (role[c] >= 0: column c absorbed column role[c]; role[r] >= 0: row r absorbed row role[r]) and
// find the nearest live neighbour.  size_old = sizes before those merges, size_new = after.
__global__ void __launch_bounds__(256) k_ward_round(float* D, int64_t ld, const int* live, int n_live, const int* role,
                                                    const float* pair_d, const int* size_old, const int* size_new,
                                                    int* nn, float* dmin) {
    const int r = live[blockIdx.x];
    const int rj = role[r];
    float* Dr = D + (int64_t)r * ld;
    const float* Dj = rj >= 0 ? D + (int64_t)rj * ld : nullptr;
    const float pdr = rj >= 0 ? pair_d[r] : 0.0f;
    const int so_r = size_old[r], sn_r = size_new[r], so_j = rj >= 0 ? size_old[rj] : 0;

    float best = __builtin_inff();
    int best_c = -1;
    for (int idx = threadIdx.x; idx < n_live; idx += 256) {
        const int c = live[idx];
        if (c == r) continue;
        const int cl = role[c];
        float v;
        if (rj < 0) {
            if (cl < 0) {
                v = Dr[c];
            } else {
                v = ward_lw(Dr[c], Dr[cl], pair_d[c], size_old[c], size_old[cl], sn_r);
                Dr[c] = v;
            }
        } else {
            if (cl < 0) {
                v = ward_lw(Dr[c], Dj[c], pdr, so_r, so_j, size_old[c]);
            } else if (r < c) {  // row merge first, then the column merge
                const float xk = ward_lw(Dr[c], Dj[c], pdr, so_r, so_j, size_old[c]);
                const float xl = ward_lw(Dr[cl], Dj[cl], pdr, so_r, so_j, size_old[cl]);
                v = ward_lw(xk, xl, pair_d[c], size_old[c], size_old[cl], sn_r);
            } else {  // mirrored entry: the same expression as row c evaluates for column r
                const float ui = ward_lw(Dr[c], Dr[cl], pair_d[c], size_old[c], size_old[cl], so_r);
                const float uj = ward_lw(Dj[c], Dj[cl], pair_d[c], size_old[c], size_old[cl], so_j);
                v = ward_lw(ui, uj, pdr, so_r, so_j, size_new[c]);
            }
            Dr[c] = v;
        }
        if (v < best) {  // c ascends within a thread: strict < keeps the lowest index on ties
            best = v;
            best_c = c;
        }
    }
    // (value, index) lexicographic minimum over the workgroup
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oc = __shfl_xor(best_c, o, 64);
        if (oc >= 0 && (best_c < 0 || ov < best || (ov == best && oc < best_c))) {
            best = ov;
            best_c = oc;
        }
    }
    __shared__ float sv[4];
    __shared__ int sc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sv[wave] = best;
        sc[wave] = best_c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sc[w] >= 0 && (best_c < 0 || sv[w] < best || (sv[w] == best && sc[w] < best_c))) {
                best = sv[w];
                best_c = sc[w];
            }
        nn[r] = best_c;
        dmin[r] = best;
    }
}

// Single workgroup (1024 threads).  Finalises the previous round's bookkeeping, detects the reciprocal
// nearest-neighbour pairs of this round in ascending slot order, logs them and compacts the live list.
__global__ void __launch_bounds__(1024) k_ward_pairs(int n, int* live, int* role, float* pair_d, int* size_old,
                                                     int* size_new, unsigned char* alive, const int* nn,
                                                     const float* dmin, int* log_i, int* log_j, float* log_d,
                                                     int* log_size, WardCounts* counts) {
    __shared__ int s_scan[1024];
    __shared__ int s_base;
    const int t = threadIdx.x;
    const int n_live = counts->n_live;
    const int m0 = counts->n_merges;
    for (int c = t; c < n; c += 1024) {
        size_old[c] = size_new[c];
        role[c] = -1;
    }
    if (t == 0) s_base = 0;
    __syncthreads();

    auto block_scan = [&](int flag) {  // exclusive prefix over the workgroup + running base
        s_scan[t] = flag;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = t >= o ? s_scan[t - o] : 0;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const int excl = s_scan[t] - flag + s_base;
        __syncthreads();
        if (t == 1023) s_base += s_scan[1023];
        __syncthreads();
        return excl;
    };

    // pairs
    for (int base = 0; base < n_live; base += 1024) {
        const int idx = base + t;
        int r = -1, c = -1, is_pair = 0;
        if (idx < n_live) {
            r = live[idx];
            c = nn[r];
            is_pair = c > r && nn[c] == r;
        }
        const int p = block_scan(is_pair);
        if (is_pair) {
            const int sz = size_old[r] + size_old[c];
            log_i[m0 + p] = r;
            log_j[m0 + p] = c;
            log_d[m0 + p] = dmin[r];
            log_size[m0 + p] = sz;
            role[r] = c;
            pair_d[r] = dmin[r];
            size_new[r] = sz;
            alive[c] = 0;
        }
    }
    __syncthreads();
    const int n_pairs = s_base;
    __syncthreads();
    if (t == 0) s_base = 0;
    __syncthreads();
    // live-list compaction in place (writes never pass the chunk being read)
    for (int base = 0; base < n_live; base += 1024) {
        const int idx = base + t;
        int r = -1, keep = 0;
        if (idx < n_live) {
            r = live[idx];
            keep = alive[r];
        }
        const int p = block_scan(keep);
        if (keep) live[p] = r;
    }
    __syncthreads();
    if (t == 0) {
        counts->n_live = s_base;
        counts->n_merges = m0 + n_pairs;
        counts->n_pairs = n_pairs;
    }
}

__global__ void __launch_bounds__(256) k_ward_init(int n, int* live, int* role, int* size_old, int* size_new,
                                                   unsigned char* alive, WardCounts* counts) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) {
        live[c] = c;
        role[c] = -1;
        size_old[c] = 1;
        size_new[c] = 1;
        alive[c] = 1;
    }
    if (c == 0) {
        counts->n_live = n;
        counts->n_merges = 0;
        counts->n_pairs = 0;
        counts->pad = 0;
    }
}

}  // namespace icv

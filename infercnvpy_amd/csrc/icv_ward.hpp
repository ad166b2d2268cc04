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

// Updated value of entry (r, c) after the merges of the previous round (role[x] >= 0: x absorbed role[x]);
// changed = the stored value has to be rewritten.  rj = role[r], cl = role[c].
struct WardRow {
    const float* Dr;
    const float* Dj;
    int r, rj, so_r, sn_r, so_j;
    float pdr;
};
__device__ __forceinline__ float ward_entry(const WardRow& R, int c, int cl, float drc, const float* pair_d,
                                            const int* size_old, const int* size_new, bool& changed) {
    changed = true;
    if (R.rj < 0) {
        if (cl < 0) {
            changed = false;
            return drc;
        }
        return ward_lw(drc, R.Dr[cl], pair_d[c], size_old[c], size_old[cl], R.sn_r);
    }
    if (cl < 0) return ward_lw(drc, R.Dj[c], R.pdr, R.so_r, R.so_j, size_old[c]);
    if (R.r < c) {  // row merge first, then the column merge
        const float xk = ward_lw(drc, R.Dj[c], R.pdr, R.so_r, R.so_j, size_old[c]);
        const float xl = ward_lw(R.Dr[cl], R.Dj[cl], R.pdr, R.so_r, R.so_j, size_old[cl]);
        return ward_lw(xk, xl, pair_d[c], size_old[c], size_old[cl], R.sn_r);
    }
    // mirrored entry: the same expression as row c evaluates for column r
    const float ui = ward_lw(drc, R.Dr[cl], pair_d[c], size_old[c], size_old[cl], R.so_r);
    const float uj = ward_lw(R.Dj[c], R.Dj[cl], pair_d[c], size_old[c], size_old[cl], R.so_j);
    return ward_lw(ui, uj, R.pdr, R.so_r, R.so_j, size_new[c]);
}

__device__ __forceinline__ void ward_argmin_publish(float best, int best_c, int r, int* nn, float* dmin) {
    // (value, index) lexicographic minimum over the 256-thread workgroup
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

// One workgroup per live row r (slot index): bring the row up to date with the merges of the previous
// round and find the nearest live neighbour.  size_old = sizes before those merges, size_new = after.
// List form: iterates the compacted live list (late rounds, few live columns).
__global__ void __launch_bounds__(256) k_ward_round(float* D, int64_t ld, const int* live, int n_live, const int* role,
                                                    const float* pair_d, const int* size_old, const int* size_new,
                                                    int* nn, float* dmin) {
    const int r = live[blockIdx.x];
    WardRow R;
    R.r = r;
    R.rj = role[r];
    float* Dr = D + (int64_t)r * ld;
    R.Dr = Dr;
    R.Dj = R.rj >= 0 ? D + (int64_t)R.rj * ld : nullptr;
    R.pdr = R.rj >= 0 ? pair_d[r] : 0.0f;
    R.so_r = size_old[r];
    R.sn_r = size_new[r];
    R.so_j = R.rj >= 0 ? size_old[R.rj] : 0;

    float best = __builtin_inff();
    int best_c = -1;
    for (int idx = threadIdx.x; idx < n_live; idx += 256) {
        const int c = live[idx];
        if (c == r) continue;
        bool changed;
        const float v = ward_entry(R, c, role[c], Dr[c], pair_d, size_old, size_new, changed);
        if (changed) Dr[c] = v;
        if (v < best) {  // c ascends within a thread: strict < keeps the lowest index on ties
            best = v;
            best_c = c;
        }
    }
    ward_argmin_publish(best, best_c, r, nn, dmin);
}

// Dense form (early rounds, most columns alive, ld % 4 == 0).  An unchanged row (the common case) is one
// contiguous float4 stream with a byte mask per four columns (bit i: column 4q + i is alive and did not merge)
// followed by the few merged columns from the round's merge list; a merged row updates every live column.
// cstate[c] = -2 dead, -1 unchanged, >= 0 the slot column c absorbed.
__global__ void __launch_bounds__(256) k_ward_round_dense(float* D, int64_t ld, int n, const int* live,
                                                          const int* cstate, const unsigned char* qmask,
                                                          const int4* mdesc, const float* mdist, int n_merged,
                                                          const float* pair_d, const int* size_old,
                                                          const int* size_new, int* nn, float* dmin) {
    const int r = live[blockIdx.x];
    WardRow R;
    R.r = r;
    R.rj = cstate[r];
    float* Dr = D + (int64_t)r * ld;
    R.Dr = Dr;
    R.Dj = R.rj >= 0 ? D + (int64_t)R.rj * ld : nullptr;
    R.pdr = R.rj >= 0 ? pair_d[r] : 0.0f;
    R.so_r = size_old[r];
    R.sn_r = size_new[r];
    R.so_j = R.rj >= 0 ? size_old[R.rj] : 0;

    float best = __builtin_inff();
    int best_c = -1;
    const int n4 = (n + 3) >> 2;  // the row stride is padded to a multiple of 4; padding columns have mask 0
    if (R.rj < 0) {
        const float4* Dr4 = reinterpret_cast<const float4*>(Dr);
        const int rq = r >> 2;
        const unsigned rbit = 1u << (r & 3);
        auto quad = [&](int q, const float4& d, unsigned m) {
            if (q == rq) m &= ~rbit;
            // ascending column order: strict < keeps the lowest index on ties
            if ((m & 1u) && d.x < best) { best = d.x; best_c = 4 * q; }
            if ((m & 2u) && d.y < best) { best = d.y; best_c = 4 * q + 1; }
            if ((m & 4u) && d.z < best) { best = d.z; best_c = 4 * q + 2; }
            if ((m & 8u) && d.w < best) { best = d.w; best_c = 4 * q + 3; }
        };
        int q = threadIdx.x;
        for (; q + 768 < n4; q += 1024) {  // four loads in flight per thread
            const float4 d0 = Dr4[q], d1 = Dr4[q + 256], d2 = Dr4[q + 512], d3 = Dr4[q + 768];
            const unsigned m0 = qmask[q], m1 = qmask[q + 256], m2 = qmask[q + 512], m3 = qmask[q + 768];
            quad(q, d0, m0);
            quad(q + 256, d1, m1);
            quad(q + 512, d2, m2);
            quad(q + 768, d3, m3);
        }
        for (; q < n4; q += 256) quad(q, Dr4[q], qmask[q]);
        // columns that merged in the previous round (Lance-Williams on this row's two entries).  The round's
        // merges are packed as {column, absorbed column, their sizes} + distance, so that the only dependent
        // accesses are the two gathers from this row; two merges per thread are in flight.
        auto merged_col = [&](const int4& m, float pd, float drc, float drl) {
            const float v = ward_lw(drc, drl, pd, m.z, m.w, R.sn_r);
            Dr[m.x] = v;
            if (v < best || (v == best && m.x < best_c)) {
                best = v;
                best_c = m.x;
            }
        };
        int idx = threadIdx.x;
        for (; idx + 256 < n_merged; idx += 512) {
            const int4 ma = mdesc[idx], mb = mdesc[idx + 256];
            const float pa = mdist[idx], pb = mdist[idx + 256];
            const float a0 = Dr[ma.x], a1 = Dr[ma.y], b0 = Dr[mb.x], b1 = Dr[mb.y];
            merged_col(ma, pa, a0, a1);
            merged_col(mb, pb, b0, b1);
        }
        for (; idx < n_merged; idx += 256) {
            const int4 ma = mdesc[idx];
            merged_col(ma, mdist[idx], Dr[ma.x], Dr[ma.y]);
        }
    } else {
        // a row that merged: every live column is updated.  Vector loads of both rows, the column states and
        // the column sizes; only columns that merged as well need gathers.
        const float4* Dr4 = reinterpret_cast<const float4*>(Dr);
        const float4* Dj4 = reinterpret_cast<const float4*>(R.Dj);
        const int4* cs4 = reinterpret_cast<const int4*>(cstate);
        const int4* so4 = reinterpret_cast<const int4*>(size_old);
        const int nq = n >> 2;
        auto elem = [&](int c, int cl, float drc, float djc, int so_c, float& out) {
            if (cl == -2 || c == r) return;
            float v;
            if (cl < 0) {
                v = ward_lw(drc, djc, R.pdr, R.so_r, R.so_j, so_c);
            } else {
                bool changed;
                v = ward_entry(R, c, cl, drc, pair_d, size_old, size_new, changed);
            }
            out = v;
            if (v < best) {
                best = v;
                best_c = c;
            }
        };
        for (int q = threadIdx.x; q < nq; q += 256) {
            const float4 d = Dr4[q], e = Dj4[q];
            const int4 cs = cs4[q], so = so4[q];
            float4 o = d;
            elem(4 * q, cs.x, d.x, e.x, so.x, o.x);
            elem(4 * q + 1, cs.y, d.y, e.y, so.y, o.y);
            elem(4 * q + 2, cs.z, d.z, e.z, so.z, o.z);
            elem(4 * q + 3, cs.w, d.w, e.w, so.w, o.w);
            reinterpret_cast<float4*>(Dr)[q] = o;
        }
        for (int c = 4 * nq + threadIdx.x; c < n; c += 256) {
            float o = Dr[c];
            elem(c, cstate[c], o, R.Dj[c], size_old[c], o);
            Dr[c] = o;
        }
    }
    ward_argmin_publish(best, best_c, r, nn, dmin);
}

// Single workgroup (1024 threads).  Finalises the previous round's bookkeeping, detects the reciprocal
// nearest-neighbour pairs of this round in ascending slot order, logs them and compacts the live list.
__global__ void __launch_bounds__(1024) k_ward_pairs(int n, int* live, int* role, int* cstate, unsigned char* qmask,
                                                     int4* mdesc, float* pair_d,
                                                     int* size_old, int* size_new, unsigned char* alive, const int* nn,
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
        cstate[c] = alive[c] ? -1 : -2;
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
            mdesc[m0 + p] = make_int4(r, c, size_old[r], size_old[c]);
            role[r] = c;
            cstate[r] = c;
            cstate[c] = -2;
            pair_d[r] = dmin[r];
            size_new[r] = sz;
            alive[c] = 0;
        }
    }
    __syncthreads();
    const int n_pairs = s_base;
    __syncthreads();
    if (t == 0) s_base = 0;
    for (int q = t; q < (n + 3) / 4; q += 1024) {  // bit i: column 4q + i takes part in the plain arg-min stream
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * q + i < n && cstate[4 * q + i] == -1) m |= 1u << i;
        qmask[q] = (unsigned char)m;
    }
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

__global__ void __launch_bounds__(256) k_ward_init(int n, int* live, int* role, int* cstate, unsigned char* qmask,
                                                   int* size_old, int* size_new, unsigned char* alive,
                                                   WardCounts* counts) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < (n + 3) / 4) {
        unsigned m = 0;
        for (int i = 0; i < 4; ++i)
            if (4 * c + i < n) m |= 1u << i;
        qmask[c] = (unsigned char)m;
    }
    if (c < n) {
        live[c] = c;
        role[c] = -1;
        cstate[c] = -1;
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

// BASELINE config 5: cell x cell squared Euclidean distances of X_cnv
// (fp32 MFMA tiles, icv_corr.hpp) and Ward linkage on the resident distance matrix.
//
// The reference has no call site for this (scanpy's heatmap dendrogram works on category means); the
// oracle is scipy: pdist(X) + linkage(y, "ward").  Ward is a reducible linkage, so every pair of
// reciprocal nearest neighbours (RNN) is a merge of the final dendrogram and all RNN pairs of one
// round can be merged at once -- the whole matrix is processed by data-parallel passes instead of
// scipy's sequential nearest-neighbour chain:
//
//   round:  k_ward_merge   one workgroup per pair merged in the previous round: the merged row (Lance-Williams
//                          in float64 on the float32 squared distances, in place), its nearest neighbour, the
//                          new column pushed into the rows that did not merge
//           k_ward_scan    one workgroup per row whose cached nearest neighbour merged or died
//           k_ward_pairs   one workgroup: detect RNN pairs in row order (deterministic), append them to
//                          the merge log, set the column states for the next round, compact the live list
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
    int n_live, n_merges, n_pairs, n_act;
};

__device__ __forceinline__ float ward_lw(float dac, float dbc, float dab, int na, int nb, int nc) {
    const double t = (double)(na + nb + nc);
    const double v = ((double)(na + nc) * (double)dac + (double)(nb + nc) * (double)dbc - (double)nc * (double)dab) / t;
    return v > 0.0 ? (float)v : 0.0f;
}

// Entry (r, c) of a row r that merged in the previous round (r absorbed rj) for a column c that merged in the
// same round (c absorbed cl): both merges are applied through one canonical order (lower slot first), so that
// row c evaluates bit for bit the same value for its column r.
struct WardRow {
    const float* Dr;
    const float* Dj;
    int r, rj, so_r, sn_r, so_j;
    float pdr;
};
__device__ __forceinline__ float ward_entry(const WardRow& R, int c, int cl, float drc, const float* pair_d,
                                            const int* size_old, const int* size_new, bool& changed) {
    changed = true;
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

// ---- storage map of a (possibly sharded) distance matrix ------------------------------------------------
// Rows are owned in super-rows of 1 << shift rows (1024, the super-tile of the distance kernel).  sr_local ==
// nullptr: one GPU holds all rows, row r is stored at row r.  Otherwise sr_local[r >> shift] is the local
// super-row index on this rank (-1: another rank's rows) and sr_global its inverse.
struct WardMap {
    const int* sr_local;
    const int* sr_global;
    int shift;
    int n_local;  // rows stored here
    __device__ __forceinline__ bool mine(int r) const { return !sr_local || sr_local[r >> shift] >= 0; }
    __device__ __forceinline__ int64_t lrow(int r) const {
        return sr_local ? ((int64_t)sr_local[r >> shift] << shift) + (r & ((1 << shift) - 1)) : (int64_t)r;
    }
    __device__ __forceinline__ int grow(int lr) const {
        return sr_global ? (sr_global[lr >> shift] << shift) + (lr & ((1 << shift) - 1)) : lr;
    }
};

// ---- round kernels ----------------------------------------------------------------------------------------
// The matrix is kept UP TO DATE at the end of every round ("push" form):
//   k_ward_merge   one workgroup per pair (i, j) merged in the previous round, run by the owner of row i: the new
//                  row D[i][c] = LW(D[i][c], D[j][c]) for every live column c (two streaming reads, one
//                  streaming write), its nearest neighbour, and -- on one GPU -- the scatter of the new value
//                  into column i of every row that did not merge (one 4-byte write per (row, merge); round 1
//                  spent two gathers and a write per (row, merge) in every live row: 3-4x the streaming time).
//                  Rows that merged in the same round are not written by others: each evaluates the shared entry
//                  itself, through one canonical order, so that the two values agree bit for bit.
//   k_ward_scatter (sharded matrices) the same column update from the new rows received from their owners.
//   k_ward_scan    nearest live neighbour of the rows whose cached neighbour merged or died.  Ward is reducible:
//                  a merged cluster is never closer to a bystander than the nearer of its parts was, so the
//                  cached nearest neighbour of every other row stays valid and the row is not read at all.
// Dense form (most columns alive, ld % 4 == 0): float4 streams over all n columns with per-column state;
// list form: gathers over the compacted live list.
// cstate[c] = -2 dead, -1 unchanged in the previous round, >= 0 the slot column c absorbed.
struct WardPairView {
    const int4* mdesc;   // {i, j, size_i, size_j} of the previous round's merges
    const float* mdist;  // their squared distances
    const int* pslot;    // staging slot of row j per merge (-1 / nullptr: row j is stored locally)
    const float* stage;
    int64_t ld_stage;
};

template <bool DENSE>
__global__ void __launch_bounds__(256) k_ward_merge(float* D, int64_t ld, int n, const int* live, int n_live,
                                                    const int* cstate, const WardPairView V, const float* pair_d,
                                                    const int* size_old, const int* size_new, const WardMap M,
                                                    bool scatter, int* nn, float* dmin) {
    const int4 m = V.mdesc[blockIdx.x];
    const int r = m.x;
    if (!M.mine(r)) return;
    WardRow R;
    R.r = r;
    R.rj = m.y;
    float* Dr = D + M.lrow(r) * ld;
    R.Dr = Dr;
    const int ps = V.pslot ? V.pslot[blockIdx.x] : -1;
    R.Dj = ps >= 0 ? V.stage + (int64_t)ps * V.ld_stage : D + M.lrow(m.y) * ld;
    R.pdr = V.mdist[blockIdx.x];
    R.so_r = m.z;
    R.so_j = m.w;
    R.sn_r = m.z + m.w;

    float best = __builtin_inff();
    int best_c = -1;
    auto elem = [&](int c, int cl, float drc, float djc, int so_c, float& out) {
        if (cl == -2 || c == r) return;
        float v;
        if (cl < 0) {
            v = ward_lw(drc, djc, R.pdr, R.so_r, R.so_j, so_c);
            // one 4-byte write per 128-byte line of row c: a read-modify-write of the line in HBM, the dominant
            // cost of the rounds (a non-temporal store was measured: 5 % slower)
            if (scatter && M.mine(c)) D[M.lrow(c) * ld + r] = v;
        } else {
            bool changed;
            v = ward_entry(R, c, cl, drc, pair_d, size_old, size_new, changed);
        }
        out = v;
        if (v < best) {  // c ascends within a thread: strict < keeps the lowest index on ties
            best = v;
            best_c = c;
        }
    };
    if (DENSE) {
        const float4* Dr4 = reinterpret_cast<const float4*>(Dr);
        const float4* Dj4 = reinterpret_cast<const float4*>(R.Dj);
        const int4* cs4 = reinterpret_cast<const int4*>(cstate);
        const int4* so4 = reinterpret_cast<const int4*>(size_old);
        const int nq = n >> 2;
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
    } else {
        for (int idx = threadIdx.x; idx < n_live; idx += 256) {
            const int c = live[idx];
            float o = Dr[c];
            const float o0 = o;
            elem(c, cstate[c], o, R.Dj[c], size_old[c], o);
            if (c != r && o != o0) Dr[c] = o;
        }
    }
    ward_argmin_publish(best, best_c, r, nn, dmin);
}

// D[c][i_q] = V[q][c] for every local row c that is alive and did not merge; V[q] = new row of merge q (the
// columns this rank owns, in local order), vrow_i[q] its slot.  grid (ceil(n_local / 256), n_v).
__global__ void __launch_bounds__(256) k_ward_scatter(float* D, int64_t ld, const float* V, int64_t ldv,
                                                      const int* vrow_i, const int* cstate, const WardMap M) {
    const int lr = blockIdx.x * 256 + threadIdx.x;
    if (lr >= M.n_local) return;
    const int c = M.grow(lr);
    if (cstate[c] != -1) return;
    D[(int64_t)lr * ld + vrow_i[blockIdx.y]] = V[(int64_t)blockIdx.y * ldv + lr];
}

// nearest live neighbour of row act[blockIdx.x] (a row that did not merge; every entry is up to date)
template <bool DENSE>
__global__ void __launch_bounds__(256) k_ward_scan(const float* D, int64_t ld, int n, const int* act, const int* live,
                                                   int n_live, const unsigned char* qmask, const WardMap M, int* nn,
                                                   float* dmin) {
    const int r = act[blockIdx.x];
    if (!M.mine(r)) return;
    const float* Dr = D + M.lrow(r) * ld;
    float best = __builtin_inff();
    int best_c = -1;
    if (DENSE) {
        const int n4 = (n + 3) >> 2;  // the row stride is padded to a multiple of 4; padding columns have mask 0
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
    } else {
        for (int idx = threadIdx.x; idx < n_live; idx += 256) {
            const int c = live[idx];
            if (c == r) continue;
            const float v = Dr[c];
            if (v < best) {
                best = v;
                best_c = c;
            }
        }
    }
    ward_argmin_publish(best, best_c, r, nn, dmin);
}

// Single workgroup (1024 threads).  Detects the reciprocal nearest-neighbour pairs of this round in ascending
// slot order, logs them, compacts the live list and lists the rows whose nearest neighbour has to be searched
// again in the next round (act: alive, did not merge, cached neighbour merged or died; all_active: every such row).
__global__ void __launch_bounds__(1024) k_ward_pairs(int n, int* live, int* cstate, unsigned char* qmask, int4* mdesc,
                                                     float* pair_d, int* size_old, int* size_new, unsigned char* alive,
                                                     const int* nn, const float* dmin, int* log_i, int* log_j,
                                                     float* log_d, int* log_size, int* act, int all_active,
                                                     WardCounts* counts) {
    __shared__ int s_scan[16];
    __shared__ int s_base;
    const int t = threadIdx.x;
    const int n_live = counts->n_live;
    const int m0 = counts->n_merges;
    for (int c = t; c < n; c += 1024) {
        size_old[c] = size_new[c];
        cstate[c] = alive[c] ? -1 : -2;
    }
    if (t == 0) s_base = 0;
    __syncthreads();

    // exclusive prefix of a 0/1 flag over the workgroup + running base: ballot / popcount inside the wavefront,
    // the 16 wavefront totals through LDS
    auto block_scan = [&](int flag) {
        const unsigned long long b = __ballot(flag != 0);
        const int lane = t & 63, wv = t >> 6;
        const int within = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_scan[wv] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = s_scan[w];
            before += w < wv ? c : 0;
            total += c;
        }
        const int excl = s_base + before + within;
        __syncthreads();
        if (t == 0) s_base += total;
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
    for (int q = t; q < (n + 3) / 4; q += 1024) {  // bit i: column 4q + i is alive
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * q + i < n && alive[4 * q + i]) m |= 1u << i;
        qmask[q] = (unsigned char)m;
    }
    __syncthreads();
    // rows to search again (before the compaction: reads the old live list)
    for (int base = 0; base < n_live; base += 1024) {
        const int idx = base + t;
        int r = -1, a = 0;
        if (idx < n_live) {
            r = live[idx];
            a = cstate[r] == -1 && (all_active || cstate[nn[r]] != -1);
        }
        const int p = block_scan(a);
        if (a) act[p] = r;
    }
    __syncthreads();
    const int n_act = s_base;
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
        counts->n_act = n_act;
    }
}

__global__ void __launch_bounds__(256) k_ward_init(int n, int* live, int* cstate, unsigned char* qmask, int* size_old,
                                                   int* size_new, unsigned char* alive, int* act, WardCounts* counts) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < (n + 3) / 4) {
        unsigned m = 0;
        for (int i = 0; i < 4; ++i)
            if (4 * c + i < n) m |= 1u << i;
        qmask[c] = (unsigned char)m;
    }
    if (c < n) {
        live[c] = c;
        act[c] = c;
        cstate[c] = -1;
        size_old[c] = 1;
        size_new[c] = 1;
        alive[c] = 1;
    }
    if (c == 0) {
        counts->n_live = n;
        counts->n_merges = 0;
        counts->n_pairs = 0;
        counts->n_act = n;
    }
}

// (nn, dmin) of the rows searched in this round, packed in list order for one all-reduce: entries of rows owned
// by other ranks are zero, so that the sum over ranks is the owner's value.  list = the round's merged rows
// (mdesc[.].x) followed by act.
__global__ void __launch_bounds__(256) k_ward_pack_nn(const int4* mdesc, int n_pairs, const int* act, int n_act,
                                                      const int* nn, const float* dmin, const WardMap M, int* out_nn,
                                                      float* out_d) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_pairs + n_act) return;
    const int r = k < n_pairs ? mdesc[k].x : act[k - n_pairs];
    const bool mine = M.mine(r);
    out_nn[k] = mine ? nn[r] : 0;
    out_d[k] = mine ? dmin[r] : 0.0f;
}
__global__ void __launch_bounds__(256) k_ward_unpack_nn(const int4* mdesc, int n_pairs, const int* act, int n_act,
                                                        const int* in_nn, const float* in_d, int* nn, float* dmin) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_pairs + n_act) return;
    const int r = k < n_pairs ? mdesc[k].x : act[k - n_pairs];
    nn[r] = in_nn[k];
    dmin[r] = in_d[k];
}

}  // namespace icv

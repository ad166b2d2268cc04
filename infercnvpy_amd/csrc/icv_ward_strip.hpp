// Ward rounds with a column layout that avoids partial-line writes ("strip" form of icv_ward.hpp's push).
//
// The in-place push writes the new distance of every bystander row c to merged cluster i into D[c][i]: one
// 4-byte write per 128-byte line of the row -- a read-modify-write of the line in HBM, measured as the dominant
// cost of the rounds (profiles/r02_config5_ward_rounds_100k.txt: 214 of 283 ms at 100 000 cells).  Here a merged
// cluster keeps its ROW (slot i) but gets a NEW COLUMN: the clusters merged in a round take the consecutive
// positions width .. width + n_pairs - 1 of a spare region of the row stride (callers allocate ld >= 1.5 n), so
// the update of a bystander row is one contiguous strip of n_pairs floats, written through an LDS transpose of
// the new rows (k_ward_push).  Columns are addressed through slot <-> position maps (replicated, deterministic);
// when the spare region is full the alive columns of the alive rows are compacted in place (k_ward_compact_*):
// once per run with ld = 1.5 n, and whenever fewer than half of the positions are alive.
// Arithmetic per entry, canonical evaluation order, tie-breaking (lowest slot) and therefore results are those of
// the in-place kernels, bit for bit.
#pragma once
#include "icv_ward.hpp"

namespace icv {

struct WardStripCounts {
    int n_live, n_merges, n_pairs, n_act, width, width_prev, need_compact, n_unmerged;
};

struct WardPos {
    int* slot_pos;          // [n]   column position of an alive slot
    int* pos_slot;          // [cap] slot whose column is (or last was) at a position; -1: never used
    unsigned char* palive;  // [cap] the position holds the column of an alive cluster
    int* pstate;            // [cap] view of the merge kernel: -2 dead, -1 alive and unchanged, >= 0 position of the absorbed partner
    int* psize;             // [cap] size before the round's merges of the cluster at the position
    int* newpos;            // [cap] compaction map (-1: dead)
};

// One workgroup per pair (i, j) merged in the previous round, run by the owner of row i: the new row over the
// positions that existed before the round (in place) and over the round's new positions (entries to clusters that
// merged in the same round), and its nearest neighbour.  mpos[p] = {old position of i, of j, new position of i}.
template <bool DENSE>
__global__ void __launch_bounds__(256) k_ward_merge_s(float* D, int64_t ld, int width_prev, const int* live, int n_live,
                                                      const int* cstate, const WardPairView V, const int4* mpos,
                                                      int n_pairs, const WardPos P, const WardMap M, int* nn,
                                                      float* dmin) {
    const int4 m = V.mdesc[blockIdx.x];
    const int r = m.x;
    if (!M.mine(r)) return;
    float* Dr = D + M.lrow(r) * ld;
    const int ps = V.pslot ? V.pslot[blockIdx.x] : -1;
    const float* Dj = ps >= 0 ? V.stage + (int64_t)ps * V.ld_stage : D + M.lrow(m.y) * ld;
    const float pdr = V.mdist[blockIdx.x];
    const int so_r = m.z, so_j = m.w, sn_r = m.z + m.w;

    float best = __builtin_inff();
    int best_c = -1;  // slot of the nearest cluster; ties go to the lowest slot, as in the in-place kernels
    auto cand = [&](float v, int q, int c) {
        if (v < best) {
            best = v;
            best_c = c < 0 ? P.pos_slot[q] : c;
        } else if (v == best) {
            const int cc = c < 0 ? P.pos_slot[q] : c;
            if (best_c < 0 || cc < best_c) best_c = cc;
        }
    };
    // columns whose cluster did not merge: in place (returns true if `out` changed)
    auto elem = [&](int q, int st, float a, float b, int sz, float& out) {
        if (st != -1) return false;  // dead, or merged in this round (second loop)
        const float v = ward_lw(a, b, pdr, so_r, so_j, sz);
        out = v;
        cand(v, q, -1);
        return true;
    };
    if (DENSE) {
        const float4* Dr4 = reinterpret_cast<const float4*>(Dr);
        const float4* Dj4 = reinterpret_cast<const float4*>(Dj);
        const int4* st4 = reinterpret_cast<const int4*>(P.pstate);
        const int4* sz4 = reinterpret_cast<const int4*>(P.psize);
        const int nq = width_prev >> 2;
        for (int q = threadIdx.x; q < nq; q += 256) {
            const float4 d = Dr4[q], e = Dj4[q];
            const int4 st = st4[q], sz = sz4[q];
            float4 o = d;
            elem(4 * q, st.x, d.x, e.x, sz.x, o.x);
            elem(4 * q + 1, st.y, d.y, e.y, sz.y, o.y);
            elem(4 * q + 2, st.z, d.z, e.z, sz.z, o.z);
            elem(4 * q + 3, st.w, d.w, e.w, sz.w, o.w);
            reinterpret_cast<float4*>(Dr)[q] = o;
        }
        for (int q = 4 * nq + threadIdx.x; q < width_prev; q += 256) {
            float o = Dr[q];
            if (elem(q, P.pstate[q], o, Dj[q], P.psize[q], o)) Dr[q] = o;
        }
    } else {
        for (int idx = threadIdx.x; idx < n_live; idx += 256) {
            const int c = live[idx];
            if (c == r || cstate[c] != -1) continue;
            const int q = P.slot_pos[c];
            float o = Dr[q];
            if (elem(q, -1, o, Dj[q], P.psize[q], o)) Dr[q] = o;
        }
    }
    // clusters that merged in the same round, from the round's merge list (old positions of both parts -> the new
    // position, consecutive over the list): one canonical order, lower slot first, so that row c evaluates bit for
    // bit the same value for its column r.  The first loop leaves these old positions untouched.
    for (int p = threadIdx.x; p < n_pairs; p += 256) {
        const int4 mc = V.mdesc[p];
        const int c = mc.x;
        if (c == r) continue;
        const int4 mq = mpos[p];
        const int q = mq.x, ql = mq.y, sz = mc.z, szl = mc.w;
        const float pdc = V.mdist[p];
        const float a = Dr[q], b = Dj[q], al = Dr[ql], bl = Dj[ql];
        float v;
        if (r < c) {
            const float xk = ward_lw(a, b, pdr, so_r, so_j, sz);
            const float xl = ward_lw(al, bl, pdr, so_r, so_j, szl);
            v = ward_lw(xk, xl, pdc, sz, szl, sn_r);
        } else {
            const float ui = ward_lw(a, al, pdc, sz, szl, so_r);
            const float uj = ward_lw(b, bl, pdc, sz, szl, so_j);
            v = ward_lw(ui, uj, pdr, so_r, so_j, sz + szl);
        }
        Dr[mq.z] = v;
        cand(v, mq.z, c);
    }
    ward_argmin_publish(best, best_c, r, nn, dmin);
}

// D[row c][width_prev + p] = D[row i_p][position of c] for the alive rows c that did not merge (ulist) and the
// merges p of the round: 64 x 64 tiles through LDS, both sides contiguous along the fast index.  One GPU.
__global__ void __launch_bounds__(256) k_ward_push(float* D, int64_t ld, int width_prev, const int4* mdesc, int n_pairs,
                                                   const int* ulist, int n_u, const int* slot_pos) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = c0 + tx < n_u ? ulist[c0 + tx] : -1;
    const int qc = c >= 0 ? slot_pos[c] : 0;
    for (int pp = ty; pp < 64; pp += 4) {
        float v = 0.0f;
        if (p0 + pp < n_pairs && c >= 0) v = D[(int64_t)mdesc[p0 + pp].x * ld + qc];
        tile[pp][tx] = v;
    }
    __syncthreads();
    if (p0 + tx < n_pairs)
        for (int cc = ty; cc < 64; cc += 4)
            if (c0 + cc < n_u) D[(int64_t)ulist[c0 + cc] * ld + width_prev + p0 + tx] = tile[tx][cc];
}

// The same strip update from exchanged rows (sharded matrices): V[vrow_of_p[p]][lr] = new distance of merge p's
// cluster to the cluster of local row lr; written for the local rows that are alive and did not merge.
__global__ void __launch_bounds__(256) k_ward_scatter_s(float* D, int64_t ld, int width_prev, const float* V, int64_t ldv,
                                                        const int* vrow_of_p, int n_pairs, const int* cstate,
                                                        const WardMap M) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int pp = ty; pp < 64; pp += 4) {
        float v = 0.0f;
        if (p0 + pp < n_pairs && r0 + tx < M.n_local) v = V[(int64_t)vrow_of_p[p0 + pp] * ldv + r0 + tx];
        tile[pp][tx] = v;
    }
    __syncthreads();
    if (p0 + tx < n_pairs)
        for (int rr = ty; rr < 64; rr += 4) {
            const int lr = r0 + rr;
            if (lr < M.n_local && cstate[M.grow(lr)] == -1) D[(int64_t)lr * ld + width_prev + p0 + tx] = tile[tx][rr];
        }
}

// out[q][k] = D[rows_l[q]][position of slots[k]]: the columns of new rows another rank needs (its local rows'
// clusters).  slot_pos == nullptr: in-place layout (position = slot).  grid (ceil(n_slots / 256), n_q).
__global__ void __launch_bounds__(256) k_ward_gather(const float* D, int64_t ld, const int64_t* rows_l, const int* slots,
                                                     int n_slots, const int* slot_pos, int limit, float* out,
                                                     int64_t ldo) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_slots) return;
    int q = slot_pos ? slot_pos[slots[k]] : slots[k];
    q = q < 0 ? 0 : (q >= limit ? limit - 1 : q);  // padding / dead slots: any valid column
    out[(int64_t)blockIdx.y * ldo + k] = D[rows_l[blockIdx.y] * ld + q];
}

// nearest live neighbour of row act[blockIdx.x] (a row that did not merge; every alive position is up to date)
template <bool DENSE>
__global__ void __launch_bounds__(256) k_ward_scan_s(const float* D, int64_t ld, int width, const int* act, const int* live,
                                                     int n_live, const unsigned char* qmask, const WardPos P,
                                                     const WardMap M, int* nn, float* dmin) {
    const int r = act[blockIdx.x];
    if (!M.mine(r)) return;
    const float* Dr = D + M.lrow(r) * ld;
    const int own_q = P.slot_pos[r];
    float best = __builtin_inff();
    int best_c = -1;  // slot; ties go to the lowest slot (looked up only when a candidate ties or wins)
    auto cand = [&](float v, int q) {
        if (v < best) {
            best = v;
            best_c = P.pos_slot[q];
        } else if (v == best) {
            const int c = P.pos_slot[q];
            if (best_c < 0 || c < best_c) best_c = c;
        }
    };
    if (DENSE) {
        const int n4 = (width + 3) >> 2;  // positions past the width have mask 0
        const float4* Dr4 = reinterpret_cast<const float4*>(Dr);
        const int rq = own_q >> 2;
        const unsigned rbit = 1u << (own_q & 3);
        auto quad = [&](int q, const float4& d, unsigned m) {
            if (q == rq) m &= ~rbit;
            if ((m & 1u) && d.x <= best) cand(d.x, 4 * q);
            if ((m & 2u) && d.y <= best) cand(d.y, 4 * q + 1);
            if ((m & 4u) && d.z <= best) cand(d.z, 4 * q + 2);
            if ((m & 8u) && d.w <= best) cand(d.w, 4 * q + 3);
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
            const float v = Dr[P.slot_pos[c]];
            if (v < best || (v == best && (best_c < 0 || c < best_c))) {
                best = v;
                best_c = c;
            }
        }
    }
    ward_argmin_publish(best, best_c, r, nn, dmin);
}

// Start of a round's bookkeeping, wide: sizes and states by slot, merge-kernel view by position.
__global__ void __launch_bounds__(256) k_ward_prep_s(int n, int* cstate, int* size_old, const int* size_new,
                                                     const unsigned char* alive, const WardPos P,
                                                     const WardStripCounts* counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        size_old[i] = size_new[i];
        cstate[i] = alive[i] ? -1 : -2;
    }
    if (i < counts->width) {
        const int a = P.palive[i];
        P.pstate[i] = a ? -1 : -2;
        P.psize[i] = a ? size_new[P.pos_slot[i]] : 0;  // sizes before this round's merges
    }
}
// bit i of qmask[q]: position 4q + i holds an alive column (after the round's merges)
__global__ void __launch_bounds__(256) k_ward_qmask_s(int cap, const unsigned char* palive, unsigned char* qmask) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= (cap + 3) / 4) return;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * q + i < cap && palive[4 * q + i]) m |= 1u << i;
    qmask[q] = (unsigned char)m;
}

// Single workgroup (1024 threads), after k_ward_prep_s: the reciprocal pairs of the round in slot order, their new
// columns, the lists for the next round (rows to search again, rows that did not merge, live rows).  Nothing is
// committed (and need_compact is set) if the round's new columns do not fit the spare region.
__global__ void __launch_bounds__(1024) k_ward_pairs_s(int n, int cap, int* live, int* cstate, int4* mdesc, int4* mpos,
                                                       float* pair_d, const int* size_old, int* size_new,
                                                       unsigned char* alive, const int* nn, const float* dmin,
                                                       int* log_i, int* log_j, float* log_d, int* log_size, int* act,
                                                       int* ulist, int all_active, const WardPos P,
                                                       WardStripCounts* counts) {
    __shared__ int s_cnt[3][16];
    __shared__ int s_base[3];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n_live = counts->n_live;
    const int m0 = counts->n_merges;
    const int W = counts->width;
    if (t < 3) s_base[t] = 0;
    __syncthreads();
    // exclusive prefixes of up to three 0/1 flags over the workgroup + running bases (ballot / popcount inside the
    // wavefront, the 16 wavefront totals through LDS)
    auto scan3 = [&](int f0, int f1, int f2, int& e0, int& e1, int& e2) {
        const unsigned long long b0 = __ballot(f0 != 0), b1 = __ballot(f1 != 0), b2 = __ballot(f2 != 0);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (lane == 0) {
            s_cnt[0][wv] = __popcll(b0);
            s_cnt[1][wv] = __popcll(b1);
            s_cnt[2][wv] = __popcll(b2);
        }
        __syncthreads();
        int bf0 = 0, bf1 = 0, bf2 = 0, t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c0 = s_cnt[0][w], c1 = s_cnt[1][w], c2 = s_cnt[2][w];
            bf0 += w < wv ? c0 : 0;
            bf1 += w < wv ? c1 : 0;
            bf2 += w < wv ? c2 : 0;
            t0 += c0;
            t1 += c1;
            t2 += c2;
        }
        e0 = s_base[0] + bf0 + __popcll(b0 & lt);
        e1 = s_base[1] + bf1 + __popcll(b1 & lt);
        e2 = s_base[2] + bf2 + __popcll(b2 & lt);
        __syncthreads();
        if (t == 0) {
            s_base[0] += t0;
            s_base[1] += t1;
            s_base[2] += t2;
        }
        __syncthreads();
    };
    auto is_pair_at = [&](int idx, int& r, int& c) {
        r = -1;
        c = -1;
        if (idx >= n_live) return 0;
        r = live[idx];
        c = nn[r];
        return (c > r && nn[c] == r) ? 1 : 0;
    };

    // pass 1: how many pairs?
    int local = 0;
    for (int idx = t; idx < n_live; idx += 1024) {
        int r, c;
        local += is_pair_at(idx, r, c);
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if (lane == 0) s_cnt[0][wv] = local;
    __syncthreads();
    int n_pairs = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) n_pairs += s_cnt[0][w];
    __syncthreads();
    if (W + n_pairs > cap) {
        if (t == 0) {
            counts->need_compact = 1;
            counts->n_pairs = 0;
        }
        return;
    }
    // pass 2: commit in slot order
    int e0, e1, e2;
    for (int base = 0; base < n_live; base += 1024) {
        int r, c;
        const int ip = is_pair_at(base + t, r, c);
        scan3(ip, 0, 0, e0, e1, e2);
        if (ip) {
            const int p = e0;
            const int sz = size_old[r] + size_old[c];
            const int qi = P.slot_pos[r], qj = P.slot_pos[c], qn = W + p;
            log_i[m0 + p] = r;
            log_j[m0 + p] = c;
            log_d[m0 + p] = dmin[r];
            log_size[m0 + p] = sz;
            mdesc[m0 + p] = make_int4(r, c, size_old[r], size_old[c]);
            mpos[m0 + p] = make_int4(qi, qj, qn, 0);
            cstate[r] = c;
            cstate[c] = -2;
            pair_d[r] = dmin[r];
            size_new[r] = sz;
            alive[c] = 0;
            P.pstate[qi] = qj;
            P.pstate[qj] = -2;
            P.slot_pos[r] = qn;
            P.pos_slot[qn] = r;
            P.palive[qi] = 0;
            P.palive[qj] = 0;
            P.palive[qn] = 1;
        }
    }
    __syncthreads();
    if (t < 3) s_base[t] = 0;
    __syncthreads();
    // pass 3: rows to search again | alive rows that did not merge | live-list compaction (in place: writes never
    // pass the chunk being read)
    for (int base = 0; base < n_live; base += 1024) {
        const int idx = base + t;
        int r = -1, fa = 0, fu = 0, fk = 0;
        if (idx < n_live) {
            r = live[idx];
            const int cs = cstate[r];
            fu = cs == -1;
            fa = fu && (all_active || cstate[nn[r]] != -1);
            fk = alive[r];
        }
        scan3(fa, fu, fk, e0, e1, e2);
        if (fa) act[e0] = r;
        if (fu) ulist[e1] = r;
        if (fk) live[e2] = r;
    }
    __syncthreads();
    if (t == 0) {
        counts->n_live = s_base[2];
        counts->n_merges = m0 + n_pairs;
        counts->n_pairs = n_pairs;
        counts->n_act = s_base[0];
        counts->width_prev = W;
        counts->width = W + n_pairs;
        counts->need_compact = 0;
        counts->n_unmerged = s_base[1];
    }
}

__global__ void __launch_bounds__(256) k_ward_init_s(int n, int cap, const WardPos P, int* ulist, unsigned char* qmask,
                                                     WardStripCounts* counts) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < (cap + 3) / 4) {
        unsigned m = 0;
        for (int i = 0; i < 4; ++i)
            if (4 * q + i < n) m |= 1u << i;
        qmask[q] = (unsigned char)m;
    }
    if (q < cap) {
        P.pos_slot[q] = q < n ? q : -1;
        P.palive[q] = q < n ? 1 : 0;
        P.pstate[q] = q < n ? -1 : -2;
        P.psize[q] = q < n ? 1 : 0;
        P.newpos[q] = -1;
    }
    if (q < n) {
        P.slot_pos[q] = q;
        ulist[q] = q;
    }
    if (q == 0) {
        counts->n_live = n;
        counts->n_merges = 0;
        counts->n_pairs = 0;
        counts->n_act = n;
        counts->width = n;
        counts->width_prev = n;
        counts->need_compact = 0;
        counts->n_unmerged = n;
    }
}

// Compaction, part 1 (one workgroup): new position of every alive position = its rank, maps rebuilt in place.
__global__ void __launch_bounds__(1024) k_ward_compact_map(int cap, const WardPos P, unsigned char* qmask,
                                                           WardStripCounts* counts) {
    __shared__ int s_scan[16];
    __shared__ int s_base;
    const int t = threadIdx.x;
    const int W = counts->width;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < W; base += 1024) {
        const int q = base + t;
        const int a = q < W ? P.palive[q] : 0;
        const int slot = a ? P.pos_slot[q] : -1;
        const unsigned long long b = __ballot(a != 0);
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
        const int np = s_base + before + within;
        __syncthreads();  // every read of this chunk precedes the writes below (targets are <= their sources)
        if (q < W) P.newpos[q] = a ? np : -1;
        if (a) {
            P.pos_slot[np] = slot;
            P.slot_pos[slot] = np;
        }
        if (t == 0) s_base += total;
        __syncthreads();
    }
    const int n_alive = s_base;
    __syncthreads();
    for (int q = t; q < cap; q += 1024) {
        P.palive[q] = q < n_alive ? 1 : 0;
        if (q >= n_alive) P.pos_slot[q] = -1;
    }
    for (int q = t; q < (cap + 3) / 4; q += 1024) {
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * q + i < n_alive) m |= 1u << i;
        qmask[q] = (unsigned char)m;
    }
    if (t == 0) {
        counts->width_prev = W;  // the row kernel moves the first width_prev positions
        counts->width = n_alive;
        counts->need_compact = 0;
    }
}

// Compaction, part 2: one workgroup per alive row (live[blockIdx.x]), in place, 4096 positions per step (16-byte
// loads of the row and of the map): all reads of a step precede its writes, and a write never passes the positions
// still to be read (newpos[q] <= q).  newpos is only valid below width_old.
__global__ void __launch_bounds__(256) k_ward_compact_rows(float* D, int64_t ld, int width_old, const int* live,
                                                           const int* newpos, const WardMap M) {
    const int r = live[blockIdx.x];
    if (!M.mine(r)) return;
    float* Dr = D + M.lrow(r) * ld;
    const float4* Dr4 = reinterpret_cast<const float4*>(Dr);
    const int4* np4 = reinterpret_cast<const int4*>(newpos);
    for (int base = 0; base < width_old; base += 4096) {
        float4 v[4];
        int4 np[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = base + 4 * (k * 256 + threadIdx.x);
            np[k] = make_int4(-1, -1, -1, -1);
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < width_old) {  // ld and the map are padded to multiples of 4: the vector loads stay inside
                np[k] = np4[q >> 2];
                v[k] = Dr4[q >> 2];
                if (q + 1 >= width_old) np[k].y = -1;
                if (q + 2 >= width_old) np[k].z = -1;
                if (q + 3 >= width_old) np[k].w = -1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (np[k].x >= 0) Dr[np[k].x] = v[k].x;
            if (np[k].y >= 0) Dr[np[k].y] = v[k].y;
            if (np[k].z >= 0) Dr[np[k].z] = v[k].z;
            if (np[k].w >= 0) Dr[np[k].w] = v[k].w;
        }
    }
}

}  // namespace icv

// The reference-order float32 column chain of a DENSE matrix, evaluated by BLOCKS of rows that do not wait for each
// other (reference tl/_infercnv.py:385, :400: np.mean(X, axis=0) of a C-contiguous matrix = per column ONE sequential
// chain s = fl(s + x_r) in float32).  k_colchain evaluates that chain at the rate HBM streams the matrix -- on ONE GPU;
// row shards on several GPUs have to take turns (dist.reference_means_chained).  This file is the form whose work does
// not serialise (prototype and proofs: tests/exact_chain_proto.py, tests/test_exact_chain_proto.py):
//
//   Inside one binade the chain is integer arithmetic.  With s = m u (u = ulp(s) = 2^(e-23), 2^23 <= m < 2^24) and
//   x >= 0: fl(s + x) = (m + q) u with q = round_half_even(x / u), and q does not depend on m unless x / u has the
//   fraction exactly 1/2 (a TIE: the parity of m decides).  For a block of rows whose chain stays inside the binade it
//   starts in and that holds no tie, the block's whole effect is m += Q, Q = sum of q: integer additions -- any order, any
//   rank, any time.  The block only needs the BINADE of its start value, which a float64 estimate of the prefix sum gives.
//
//   k_chain_records  (concurrent over blocks, slabs, ranks): per (64-row block, column) the record {exponent assumed, Q}
//                    from the float64 estimate of the block's start; blocks that cannot be summarised -- a tie, a
//                    negative / non-finite entry, no normal start yet, Q leaving the binade by itself, the estimate
//                    within kBkMargin ulps of the binade's end -- get their 64 values copied to a stash instead; a slab
//                    of 16 blocks that agree on the exponent and hold no such block also gets ONE record.
//   k_chain_scan     (in row order, but only over the records): per column s_bits += Q where the record is valid for the
//                    TRUE start (exponent equal, mantissa + Q below 2^24: an integer add on the float's bit pattern), a
//                    sequential replay of the 64 stashed values (or of the matrix rows) elsewhere.
//
// A record is 4 bytes per 64 x 4 bytes of matrix; the scan reads one slab record per 1024 rows and column.  Columns with
// negative entries, NaNs or ties lose nothing but time (their blocks are replayed); float64 matrices, CSR and row lists keep
// the chain kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace icv {

constexpr int kBkRows = 64;                       // rows per block
constexpr int kBkPerSlab = 16;                    // blocks per slab
constexpr int kBkSlab = kBkRows * kBkPerSlab;     // 1024 rows: the slab of k_colsum_dense's partial sums
constexpr unsigned kBkMargin = 1u << 13;          // ulps: the float64 estimate may be this far from the float32 chain
constexpr unsigned kBkReplay = 0x80000000u;       // record: replay; low 31 bits = stash slot, kBkNoSlot = from the matrix
constexpr unsigned kBkNoSlot = 0x7fffffffu;

// slab_start[s][c] = est_start[c] + sum of partial[s'][c], s' < s (in place over `partial`); total[c] = the sum of all
__global__ void __launch_bounds__(256) k_blocks_prefix(double* __restrict__ partial, int n_slabs, int n_cols,
                                                       const double* __restrict__ est_start, double* __restrict__ total) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n_cols) return;
    double run = est_start ? est_start[col] : 0.0, tot = 0.0;
    for (int s = 0; s < n_slabs; ++s) {
        const double p = partial[(int64_t)s * n_cols + col];
        partial[(int64_t)s * n_cols + col] = run;
        run += p;
        tot += p;
    }
    if (total) total[col] = tot;
}

// one workgroup = 256 columns x one slab of 1024 rows; a thread walks its column block by block
__global__ void __launch_bounds__(256) k_chain_records(const float* __restrict__ x, int64_t n_rows, int64_t ld, int n_cols,
                                                       const double* __restrict__ slab_start, uint32_t* __restrict__ rec,
                                                       uint32_t* __restrict__ slab_rec, float* __restrict__ stash,
                                                       unsigned* __restrict__ stash_count, unsigned stash_cap) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int64_t slab = blockIdx.y;
    const int64_t r_begin = slab * kBkSlab;
    if (col >= n_cols) return;
    double est = slab_start[slab * n_cols + col];
    unsigned eb0 = 0, q_slab = 0;
    bool slab_ok = true;
    for (int blk = 0; blk < kBkPerSlab; ++blk) {
        const int64_t r0 = r_begin + (int64_t)blk * kBkRows;
        if (r0 >= n_rows) break;
        const int nr = (int)(n_rows - r0 < kBkRows ? n_rows - r0 : kBkRows);
        const float s_est = (float)est;
        const unsigned sb = __float_as_uint(s_est);
        const unsigned eb = sb >> 23;  // biased exponent (a negative or NaN estimate: >= 256, bad below)
        // 2^(23 - e) as a float needs 23 <= eb <= 254; below, the chain has not reached a normal value worth summarising
        bool bad = !(eb >= 23u && eb <= 254u);
        const float inv_u = __uint_as_float((277u - (bad ? 127u : eb)) << 23);
        unsigned q = 0;
        bool tie = false, odd = false;
        const float* xp = x + r0 * ld + col;
        if (nr == kBkRows) {
#pragma unroll 2
            for (int r = 0; r < kBkRows; r += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(xp + (int64_t)(r + u) * ld);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    est += (double)v[u];
                    const float t = v[u] * inv_u;  // exact (a power of two) unless it overflows: then `odd`
                    const float k = floorf(t), f = t - k;
                    tie |= f == 0.5f;
                    const float qf = k + (f > 0.5f ? 1.0f : 0.0f);
                    odd |= !(v[u] >= 0.0f) | !(qf < 8388608.0f);  // negative, NaN, or a quotient that leaves the binade alone
                    q += (unsigned)(int)(qf < 8388608.0f ? qf : 0.0f);
                }
            }
        } else {
            for (int r = 0; r < nr; ++r) {
                const float v = xp[(int64_t)r * ld];
                est += (double)v;
                const float t = v * inv_u;
                const float k = floorf(t), f = t - k;
                tie |= f == 0.5f;
                const float qf = k + (f > 0.5f ? 1.0f : 0.0f);
                odd |= !(v >= 0.0f) | !(qf < 8388608.0f);
                q += (unsigned)(int)(qf < 8388608.0f ? qf : 0.0f);
            }
        }
        bad |= tie | odd | (q >= 0x800000u);
        // the estimate's mantissa + Q close to the end of the binade: the true chain may cross inside this block
        bad |= (sb & 0x7fffffu) + q + kBkMargin >= 0x800000u;
        bad |= (sb & 0x7fffffu) < kBkMargin;  // ... or may still be in the binade below at the block's start
        unsigned record;
        if (bad) {
            unsigned slot = kBkNoSlot;
            if (stash) {
                const unsigned got = atomicAdd(stash_count, 1u);
                if (got < stash_cap) {
                    slot = got;
                    float* sp = stash + (size_t)slot * kBkRows;
                    for (int r = 0; r < kBkRows; ++r) sp[r] = r < nr ? xp[(int64_t)r * ld] : 0.0f;  // (L2 hits)
                }
            }
            record = kBkReplay | slot;
            slab_ok = false;
        } else {
            record = (eb << 23) | q;
            if (blk == 0) eb0 = eb;
            slab_ok = slab_ok && eb == eb0;
            q_slab += q;
        }
        rec[(slab * kBkPerSlab + blk) * (int64_t)n_cols + col] = record;
    }
    // the slab as ONE record where its blocks agree (the scan tests mantissa + Q against the binade's end itself)
    slab_rec[slab * n_cols + col] = (slab_ok && q_slab < 0x800000u) ? ((eb0 << 23) | q_slab) : (kBkReplay | kBkNoSlot);
}

// acc[c] (the exact float32 chain values before row 0 of this matrix, in / out): one thread per column walks the slab
// records, the block records of slabs without one, and replays what has to be replayed
__global__ void __launch_bounds__(256) k_chain_scan(const float* __restrict__ x, int64_t n_rows, int64_t ld, int n_cols,
                                                    int64_t rec_ld /* columns of the record tables (n_cols of a column range may be fewer) */,
                                                    const uint32_t* __restrict__ rec, const uint32_t* __restrict__ slab_rec,
                                                    const float* __restrict__ stash, float* __restrict__ acc,
                                                    unsigned long long* __restrict__ n_replayed) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n_cols) return;
    const int64_t n_slabs = (n_rows + kBkSlab - 1) / kBkSlab;
    unsigned sb = __float_as_uint(acc[col]);
    unsigned replays = 0;
    const auto apply = [&](unsigned r) -> bool {  // a summarising record: valid for the true start?
        const unsigned eb = (r >> 23) & 0xffu, q = r & 0x7fffffu;
        if (!(r & kBkReplay) && (sb >> 23) == eb && (sb & 0x7fffffu) + q < 0x800000u) {
            sb += q;  // the mantissa grows by Q ulps: an integer add on the bit pattern
            return true;
        }
        return false;
    };
    constexpr int PFS = 8;  // slab records in flight
    for (int64_t s0 = 0; s0 < n_slabs; s0 += PFS) {
        uint32_t sr[PFS];
#pragma unroll
        for (int u = 0; u < PFS; ++u) sr[u] = s0 + u < n_slabs ? slab_rec[(s0 + u) * rec_ld + col] : 0u;
#pragma unroll 1
        for (int u = 0; u < PFS && s0 + u < n_slabs; ++u) {
            if (apply(sr[u])) continue;
            const int64_t s = s0 + u;
            uint32_t br[kBkPerSlab];
#pragma unroll
            for (int b = 0; b < kBkPerSlab; ++b) {
                const int64_t r0 = (s * kBkPerSlab + b) * kBkRows;
                br[b] = r0 < n_rows ? rec[(s * kBkPerSlab + b) * rec_ld + col] : 0u;  // (0: adds nothing)
            }
#pragma unroll 1
            for (int b = 0; b < kBkPerSlab; ++b) {
                const int64_t r0 = (s * kBkPerSlab + b) * kBkRows;
                if (r0 >= n_rows) break;
                if (apply(br[b])) continue;
                // replay: the block's rows one after the other, in float32 (what the chain kernels do for every row)
                ++replays;
                float sv = __uint_as_float(sb);
                const unsigned slot = br[b] & 0x7fffffffu;
                if ((br[b] & kBkReplay) && slot != kBkNoSlot && stash) {
                    const float4* sp = reinterpret_cast<const float4*>(stash + (size_t)slot * kBkRows);
                    float4 v[kBkRows / 4];
#pragma unroll
                    for (int i = 0; i < kBkRows / 4; ++i) v[i] = sp[i];
#pragma unroll
                    for (int i = 0; i < kBkRows / 4; ++i) {
                        sv = sv + v[i].x;
                        sv = sv + v[i].y;
                        sv = sv + v[i].z;
                        sv = sv + v[i].w;
                    }
                } else {
                    const int nr = (int)(n_rows - r0 < kBkRows ? n_rows - r0 : kBkRows);
                    const float* xp = x + r0 * ld + col;
                    for (int r = 0; r < nr; ++r) sv = sv + xp[(int64_t)r * ld];
                }
                sb = __float_as_uint(sv);
            }
        }
    }
    acc[col] = __uint_as_float(sb);
    if (n_replayed && replays) atomicAdd(n_replayed, (unsigned long long)replays);
}

}  // namespace icv

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

#include "icv_kernels.hpp"

namespace icv {

constexpr int kBkRows = 64;                       // rows per block
constexpr int kBkPerSlab = 16;                    // blocks per slab
constexpr int kBkSlab = kBkRows * kBkPerSlab;     // 1024 rows: the slab of k_colsum_dense's partial sums
constexpr unsigned kBkMargin = 1u << 13;          // ulps: the float64 estimate may be this far from the float32 chain
constexpr unsigned kBkReplay = 0x80000000u;       // record: replay; low 31 bits = stash slot, kBkNoSlot = from the matrix
constexpr unsigned kBkNoSlot = 0x7fffffffu;

// slab_start[s][c] = est_start[c] + sum of partial[s'][c], s' < s
__global__ void __launch_bounds__(256) k_blocks_prefix(const double* __restrict__ partial, int n_slabs, int n_cols,
                                                       const double* __restrict__ est_start, double* __restrict__ slab_start) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n_cols) return;
    double run = est_start ? est_start[col] : 0.0;
    for (int s = 0; s < n_slabs; ++s) {
        slab_start[(int64_t)s * n_cols + col] = run;
        run += partial[(int64_t)s * n_cols + col];
    }
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
                float v[8];  // (16 loads in flight measured the same: 2.0 ms per 125 000 x 20 000)
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

// acc[c] (the exact float32 chain values before row 0 of this matrix, in / out).  ONE WAVEFRONT PER COLUMN: the lanes
// hold 64 consecutive slab records, a DPP prefix sum applies every run of valid ones at once (valid for the TRUE start:
// exponent equal, mantissa + prefix below 2^24 -- an integer add on the bit pattern), the first invalid slab is opened
// (its 16 block records, the same way) and a block that has to be replayed arrives as ONE 256-byte load of its 64
// stashed values (or 64 strided matrix loads) that the wavefront adds in order through lane reads.  (First version: one
// THREAD per column -- the ~5 replays per column and rank, each a dependent load + 64 adds, ran one after the other
// for the 64 columns of a wavefront: 0.28 ms per rank, serial over the ranks; this form: the latency of a replay is
// paid per column, 8 192 wavefronts at a time.)
__global__ void __launch_bounds__(256) k_chain_scan(const float* __restrict__ x, int64_t n_rows, int64_t ld, int n_cols,
                                                    int64_t rec_ld /* columns of the record tables (n_cols of a column range may be fewer) */,
                                                    const uint32_t* __restrict__ rec, const uint32_t* __restrict__ slab_rec,
                                                    const float* __restrict__ stash, float* __restrict__ acc,
                                                    unsigned long long* __restrict__ n_replayed) {
    const int lane = threadIdx.x & 63;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const int64_t n_slabs = (n_rows + kBkSlab - 1) / kBkSlab;
    unsigned replays = 0;
    for (int col = wave_id; col < n_cols; col += n_waves) {
        unsigned sb = __float_as_uint(acc[col]);  // (uniform)
        // records r[0 .. cnt) in the lanes: apply every valid run, call `open(i)` for the first record that is not
        const auto walk = [&](unsigned r, int cnt, auto&& open) {
            int pos = 0;
            while (pos < cnt) {
                const bool live = lane >= pos && lane < cnt;
                const bool plain = live && !(r & kBkReplay);
                const int q = plain ? (int)(r & 0x7fffffu) : 0;
                const int incl = wave_scan_dpp(q);
                const bool ok = lane < pos || (plain && ((r >> 23) & 0xffu) == (sb >> 23) &&
                                               (sb & 0x7fffffu) + (unsigned)incl < 0x800000u);
                const unsigned long long bad = ~__builtin_amdgcn_ballot_w64(ok);
                int first = bad ? (int)__builtin_ctzll(bad) : 64;
                first = first < cnt ? first : cnt;
                if (first > pos) sb += (unsigned)__builtin_amdgcn_readlane(incl, first - 1);
                if (first >= cnt) break;
                open(first);
                pos = first + 1;
            }
        };
        for (int64_t s0 = 0; s0 < n_slabs; s0 += 64) {
            const int cnt = (int)(n_slabs - s0 < 64 ? n_slabs - s0 : 64);
            const unsigned sr = lane < cnt ? slab_rec[(s0 + lane) * rec_ld + col] : 0u;
            walk(sr, cnt, [&](int si) {
                const int64_t s = s0 + si;
                const int64_t rows_left = n_rows - s * kBkSlab;
                const int nb = (int)(rows_left >= kBkSlab ? kBkPerSlab : (rows_left + kBkRows - 1) / kBkRows);
                const unsigned br = lane < nb ? rec[(s * kBkPerSlab + lane) * rec_ld + col] : 0u;
                walk(br, nb, [&](int bi) {
                    // replay: the block's rows one after the other, in float32 (what the chain kernels do for every row)
                    ++replays;
                    const unsigned rb = (unsigned)__builtin_amdgcn_readlane((int)br, bi);
                    const int64_t r0 = (s * kBkPerSlab + bi) * (int64_t)kBkRows;
                    const int nr = (int)(n_rows - r0 < kBkRows ? n_rows - r0 : kBkRows);
                    const unsigned slot = rb & 0x7fffffffu;
                    float v;
                    if ((rb & kBkReplay) && slot != kBkNoSlot && stash) v = stash[(size_t)slot * kBkRows + lane];
                    else v = lane < nr ? x[(r0 + lane) * ld + col] : 0.0f;
                    float sv = __uint_as_float(sb);
#pragma unroll 8
                    for (int i = 0; i < kBkRows; ++i)
                        sv = sv + __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), i));
                    sb = __float_as_uint(sv);
                });
            });
        }
        if (lane == 0) acc[col] = __uint_as_float(sb);
    }
    if (n_replayed && replays && lane == 0) atomicAdd(n_replayed, (unsigned long long)replays);
}

}  // namespace icv

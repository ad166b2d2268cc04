// Reference-order column sums of a CSR matrix whose work FOLLOWS THE STORED ENTRIES (reference tl/_infercnv.py:385, :400:
// scipy's `(X * (1/n)).sum(axis=0)` = per column one sequential chain acc = fl(acc + fl(x * fl(1/n))) over the rows).
//
// k_colchain_csr (icv_kernel_chain.hpp, round 4) rebuilt 60-row slices of a column tile DENSELY in LDS and let the chain
// wavefront add every LDS row: 37 LDS rows per 60 input rows at 7 % density whatever the merging, and 704 uncoalesced
// line accesses of the CU's texture path per round (lane = row: every 16-byte load of a wavefront touched 64 lines).
// Here a round of 64 input rows becomes PER-COLUMN QUEUES:
//
//   * rank of an entry (row r, column c) inside its round = number of earlier rows of the round that store column c.
//     The producer ORs bit r into a 64-bit row mask per column (LDS `ds_or_b64`: commutative, so the landing order of the
//     atomics does not matter), then rank = popcount(mask[c] & (bit(r) - 1)).  Deterministic by construction.
//   * LDS row j of the round holds the j-th stored entry of every column (zero where a column has fewer): adding a zero
//     is exact, so the chain's sums are the reference's -- and the round needs max_c count(c) LDS rows, ~11 instead of 64
//     at 7 % density (a column stored in every row degrades gracefully to the dense chain: 64 rows).
//   * the LDS rows of all rounds form ONE continuous stream in a ring (160 KB: ~380 rows of 384 B): producers allocate
//     their rows in round order (a two-word hand-over), fill them, publish in round order; the chain wavefront adds
//     whatever is published in chunks of ten rows, never looking at round boundaries.  ~30 rounds are in flight.
//   * loads: lane = (row, piece): four lanes share a row's run of the tile's entries (16 bytes of indices / values each),
//     so a load instruction touches ~16-20 lines instead of 64; the rows' bounds come from a TILE-MAJOR table of 16-bit
//     offsets (blocks of 32 rows: the 64 rows of a round read two 64-byte runs per tile boundary) written by
//     k_csr_tile_bounds16 through an LDS transpose (the table is half the size of round 4's and its lines are written whole).
//
// Needs n_cols <= 65535 (16-bit offsets inside a row); wider matrices keep the round-4 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "icv_kernel_chain.hpp"

namespace icv {

constexpr int kQRows = 64;      // input rows per round at most (one bit each in a column's row mask)
// PIECES (template parameter of k_colchain_csrq): lanes that share a row's run of the tile's entries, four entries each =
// 4 PIECES entry slots per row.  A load vector covers 64 / PIECES rows; a round is 4 vectors (2 for PIECES = 2):
//   PIECES  2: 64 rows x  8 slots (2 vectors: half the slot work -- matrices with fewer than ~2 entries per row and tile)
//   PIECES  4: 64 rows x 16 slots (round 5's shape: config 4 has 5.5 entries per row and tile)
//   PIECES  8: 32 rows x 32 slots, PIECES 16: 16 rows x 64 slots (denser rows: round 5 sent every row with more than 16
//              entries in the tile through per-entry guarded loads -- 62 % of the rounds at 14 % density, 8.0 ms against
//              3.1 at 7 %: super-linear; here the time follows the entries)
// The launcher picks PIECES from the mean and spread of the entries per (row, tile); rows beyond the slots still take
// the guarded loads (rare by that choice), and any PIECES gives the same bits.
__host__ __device__ constexpr int q_rows_per_round(int pieces) { return pieces <= 4 ? 64 : 256 / pieces; }
constexpr int kQTabRows = 32;   // rows per block of the bounds table
constexpr int kQCtl = 256;      // control words at the end of the LDS
constexpr int kQCmBytes = 1536; // row masks of one producer wavefront: 128 columns x 8 bytes + one trash word PER LANE
                                // (lanes without an entry OR into theirs: no branches around the atomics, and no two
                                // lanes on one address -- a single shared trash word serialised 37 lanes per atomic:
                                // 6.1 ms instead of 2.6)
constexpr int kQTrash = 512;    // LDS bytes the lanes without an entry scatter into (8 per lane)
constexpr int kQFar = 1 << 28;  // entries a buffer offset can span (far_limit); rows further apart take the guarded loads

__host__ __device__ inline int64_t q_tab_index(int64_t i, int t, int n_tiles) {
    return ((i / kQTabRows) * (int64_t)(n_tiles + 1) + t) * kQTabRows + (i % kQTabRows);
}

// tab[q_tab_index(i, t)] = number of entries of selected row i in the tiles before t (t = 0 .. n_tiles), 16 bits each.
// One workgroup (four wavefronts) per block of 32 rows; a wavefront takes one row at a time (lane = entry, 256 entries in
// flight); the block's table is collected in LDS (row-major, odd stride) and written out tile-major in whole lines.
template <bool LIST>
__global__ void __launch_bounds__(256) k_csr_tile_bounds16(const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const int32_t* __restrict__ sel, int64_t n_sel,
                                                           const uint16_t* __restrict__ line_tile, int n_lines,
                                                           int esz_shift, int n_tiles, uint32_t* __restrict__ tab32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char q_smem[];
    uint16_t* lt = reinterpret_cast<uint16_t*>(q_smem);
    const int stride = (n_tiles + 2) | 1;  // 16-bit elements per row: odd, so the transposed reads spread over the banks
    uint16_t* l_tile = lt + kQTabRows * stride;  // line -> tile, copied to LDS (a dependent global load per chunk otherwise)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int l = threadIdx.x; l < n_lines; l += 256) l_tile[l] = line_tile[l];
    // this wavefront's rows: wave, wave + 4, ... (eight of them); lane j < 8 holds the entry range of row j
    int64_t my_e0 = 0, my_len = -1;  // (-1: row past the end)
    if (lane < kQTabRows / 4) {
        const int64_t i = (int64_t)blockIdx.x * kQTabRows + wave + 4 * lane;
        if (i < n_sel) {
            const int64_t row = LIST ? sel[i] : i;
            my_e0 = indptr[row];
            my_len = indptr[row + 1] - my_e0;
        }
    }
    __syncthreads();
    const auto row_of = [&](int j, int64_t& e0, int64_t& len) {
        e0 = ((int64_t)__builtin_amdgcn_readlane((int)(my_e0 >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)my_e0, j);
        len = ((int64_t)__builtin_amdgcn_readlane((int)(my_len >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)my_len, j);
    };
    // chunks of 256 entries (one row at a time, the terminator position `len` included); the column indices of the NEXT
    // chunk -- of this row or of the wavefront's next row -- are in flight while the current chunk is processed
    int col_next[4];
    const auto load_chunk = [&](int64_t e0, int64_t len, int64_t base) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t pos = base + u * 64 + lane;
            col_next[u] = pos < len ? indices[e0 + pos] : 0;
        }
    };
    int j = 0, jn = 0;
    int64_t e0 = 0, len = -1, base = 0, e0n = 0, lenn = -1, basen = 0;
    row_of(0, e0n, lenn);
    if (lenn >= 0) load_chunk(e0n, lenn, 0);
    int carry = -1;  // tile of the entry before this chunk
    for (;;) {
        j = jn, e0 = e0n, len = lenn, base = basen;
        if (j >= kQTabRows / 4) break;
        uint16_t* row_t = lt + (wave + 4 * j) * stride;
        if (len < 0) {  // rows past the end: zeros, never read
            for (int t = lane; t <= n_tiles; t += 64) row_t[t] = 0;
            jn = j + 1;
            if (jn < kQTabRows / 4) row_of(jn, e0n, lenn);
            basen = 0;
            if (jn < kQTabRows / 4 && lenn >= 0) load_chunk(e0n, lenn, 0);
            carry = -1;
            continue;
        }
        int col[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) col[u] = col_next[u];
        // the next chunk
        basen = base + 256;
        if (basen > len) {
            jn = j + 1;
            basen = 0;
            if (jn < kQTabRows / 4) row_of(jn, e0n, lenn);
        }
        if (jn < kQTabRows / 4 && lenn >= 0) load_chunk(e0n, lenn, basen);
        int T[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t pos = base + u * 64 + lane;
            // pos == len: the terminator closes every remaining tile at `len`
            T[u] = pos < len ? (int)l_tile[((unsigned)col[u] << esz_shift) >> 7] : n_tiles;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t pos = base + u * 64 + lane;
            int P = __shfl_up(T[u], 1);
            if (lane == 0) P = carry;
            carry = __shfl(T[u], 63);
            if (pos <= len)
                for (int t = P + 1; t <= T[u]; ++t) row_t[t] = (uint16_t)pos;
        }
        if (jn != j) carry = -1;
    }
    __syncthreads();
    // tile-major: word w of the block = rows 2q, 2q + 1 of tile boundary t (w = 16 t + q)
    const int n_words = (n_tiles + 1) * (kQTabRows / 2);
    uint32_t* out = tab32 + (int64_t)blockIdx.x * n_words;
    for (int w = threadIdx.x; w < n_words; w += 256) {
        const int t = w >> 4, q = w & 15;
        out[w] = (uint32_t)lt[(2 * q) * stride + t] | ((uint32_t)lt[(2 * q + 1) * stride + t] << 16);
    }
}

// max over the 64 lanes of a non-negative int, DPP only (six dependent VALU steps instead of six LDS round trips)
__device__ __forceinline__ int wave_max_nonneg_dpp(int v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:8
    t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;  // row_bcast:15 -> rows 1, 3
    t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;  // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// what a producer does between two looks at a hand-over word (ICV_Q_SLEEP: developer knob of tools/build_variant.sh)
#ifndef ICV_Q_SLEEP
#define ICV_Q_SLEEP 1
#endif
#if ICV_Q_SLEEP > 0
#define ICV_Q_WAIT() __builtin_amdgcn_s_sleep(ICV_Q_SLEEP)
#else
#define ICV_Q_WAIT() ((void)0)
#endif
__device__ __forceinline__ unsigned q_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void q_store(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The chain wavefront: adds the published LDS rows of the ring in order, ten at a time (the hand-scheduled chunk of
// k_colchain), whatever round they belong to.  ctl: [3] rounds published, [4] rows published, [5] rows consumed.
template <typename T, int NL>
__device__ __forceinline__ void chain_stream(const unsigned char* smem, int n_ring, unsigned n_rounds, int lane,
                                             typename ChainLane<T>::type& a, unsigned* ctl) {
    typedef typename ChainLane<T>::type lane_t;
    constexpr int RB = 128 * NL;
    const unsigned lds0 = (unsigned)(uintptr_t)smem + (unsigned)lane * 8u;
    unsigned done = 0;  // rows consumed (wraps with the counters)
    int pos = 0;        // ring position of row `done`: a multiple of ten
    for (;;) {
        const unsigned rr = q_load(ctl + 3);
        const unsigned rrow = q_load(ctl + 4);
        int avail = (int)(rrow - done);
        if (avail >= 10) {
            int n_chunks = avail / 10;
            const int to_end = (n_ring - pos) / 10;
            if (n_chunks > to_end) n_chunks = to_end;
            asm volatile("" ::: "memory");  // rows written by other wavefronts: read them now
            const unsigned p0 = lds0 + (unsigned)pos * (unsigned)RB;
            lane_t va[10], vb[10];
#define ICV_Q_BODY(RBS, OP)                                                \
    ICV_CH_LOAD(RBS, va, p0);                                              \
    int c = 0;                                                             \
    for (; c + 2 < n_chunks; c += 2) {                                     \
        const unsigned p1 = p0 + (unsigned)((c + 1) * 10 * RB);            \
        ICV_CH_STEP(RBS, OP, a, vb, va, p1);                               \
        const unsigned p2 = p0 + (unsigned)((c + 2) * 10 * RB);            \
        ICV_CH_STEP(RBS, OP, a, va, vb, p2);                               \
    }                                                                      \
    if (c + 1 < n_chunks) {                                                \
        const unsigned p1 = p0 + (unsigned)((c + 1) * 10 * RB);            \
        ICV_CH_STEP(RBS, OP, a, vb, va, p1);                               \
        ICV_CH_ADDS(OP, a, vb);                                            \
    } else {                                                               \
        ICV_CH_ADDS(OP, a, va);                                            \
    }
            if constexpr (sizeof(T) == 4) {
                if constexpr (NL == 1) { ICV_Q_BODY("128", "v_pk_add_f32") }
                else if constexpr (NL == 2) { ICV_Q_BODY("256", "v_pk_add_f32") }
                else if constexpr (NL == 3) { ICV_Q_BODY("384", "v_pk_add_f32") }
                else { ICV_Q_BODY("512", "v_pk_add_f32") }
            } else {
                if constexpr (NL == 1) { ICV_Q_BODY("128", "v_add_f64") }
                else if constexpr (NL == 2) { ICV_Q_BODY("256", "v_add_f64") }
                else if constexpr (NL == 3) { ICV_Q_BODY("384", "v_add_f64") }
                else { ICV_Q_BODY("512", "v_add_f64") }
            }
#undef ICV_Q_BODY
            done += 10u * (unsigned)n_chunks;
            pos += 10 * n_chunks;
            if (pos >= n_ring) pos = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads have returned: the rows may be reused)
            if (lane == 0) q_store(ctl + 5, done);
        } else if (rr == n_rounds) {
            // every round is published: what q_load(ctl + 4) returns now is final
            const unsigned last = q_load(ctl + 4);
            avail = (int)(last - done);
            if (avail >= 10) continue;
            asm volatile("" ::: "memory");
            const lane_t* rows = reinterpret_cast<const lane_t*>(smem + (size_t)pos * RB) + lane;
            for (int i = 0; i < avail; ++i) chain_add(a, rows[i * (RB / 8)]);
            break;
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// acc[c] += fl(x * scale) over the stored entries of rows sel[0..n_sel) (nullptr: rows 0..n_sel), rows ascending.
// grid = the column tiles of ChainLaunch (= n_tiles of the table), 1024 threads: 15 producer wavefronts + the chain.
template <typename T, bool LIST, int PIECES = 4>
__global__ void __launch_bounds__(kChThreads) k_colchain_csrq(const T* __restrict__ vals,
                                                              const int64_t* __restrict__ indptr,
                                                              const int32_t* __restrict__ indices, int64_t n_rows_all,
                                                              const int32_t* __restrict__ sel, int64_t n_sel, int n_cols,
                                                              int n_lines, int lds_bytes,
                                                              const uint16_t* __restrict__ tab, T scale,
                                                              int far_limit, T* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename ChainLane<T>::type lane_t;
    constexpr int CPL = 8 / (int)sizeof(T);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // XCD x takes a contiguous eighth of the tiles (as k_colchain_csr): the runs its workgroups gather from a row's
    // entry list share their cache lines in ONE L2
    const int n_tiles = gridDim.x;
    const int xcd = blockIdx.x & 7, q_in_xcd = blockIdx.x >> 3;
    const int tile = xcd * (n_tiles >> 3) + (xcd < (n_tiles & 7) ? xcd : (n_tiles & 7)) + q_in_xcd;
    const int line0 = (int)((int64_t)tile * n_lines / n_tiles);
    const int nl = (int)((int64_t)(tile + 1) * n_lines / n_tiles) - line0;
    const int c0 = line0 * (128 / (int)sizeof(T));
    const int row_bytes = 128 * nl;
    const int n_tcols = row_bytes / (int)sizeof(T);  // columns of the tile (<= 128)
    // LDS: [ring: n_ring rows][trash bytes of the scatter: 8 per lane][row masks + trash words: 15 x 1.5 KB][control words]
    const int n_ring = (lds_bytes - kChLoaders * kQCmBytes - kQCtl - kQTrash) / row_bytes / 10 * 10;
    unsigned char* cm_base = smem + (lds_bytes - kQCtl - kChLoaders * kQCmBytes);
    // ctl: [0] next round to allocate, [1] rows allocated (wrapping counter), [2] ring position of the next row,
    //      [3] rounds published, [4] rows published, [5] rows consumed
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + lds_bytes - kQCtl);
    static_assert(PIECES == 2 || PIECES == 4 || PIECES == 8 || PIECES == 16, "lanes per row");
    constexpr int ROWS = q_rows_per_round(PIECES);  // input rows per round
    constexpr int RPV = 64 / PIECES;                 // rows per load vector
    constexpr int NVEC = ROWS / RPV;                 // load vectors per round
    constexpr int SLOTS = 4 * PIECES;                // entry slots per row
    const unsigned n_rounds = (unsigned)((n_sel + ROWS - 1) / ROWS);
    for (int o = threadIdx.x * 16; o < kChLoaders * kQCmBytes + kQCtl; o += kChThreads * 16)
        *reinterpret_cast<uint4*>(cm_base + o) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    if (wave < kChLoaders) {
        unsigned long long* cm = reinterpret_cast<unsigned long long*>(cm_base + wave * kQCmBytes);
        const int64_t e_end = indptr[n_rows_all];
        const int piece = lane & (PIECES - 1), sub = lane / PIECES;  // lane = (row RPV v + sub of the round, piece) in load vector v
        // stage A (lane = row of the round): the three loaded words are kept RAW until the next turn (arithmetic on
        // them here would make the compiler wait for the loads issued just before)
        int64_t a_rp = 0;
        unsigned a_lo = 0, a_hi = 0;
        // stage B: lane = row: where the row's entries of the tile start, how many; lane = (row, piece): four entries
        int64_t b_base = 0;
        int b_cnt = 0;
        bool b_more = false;  // (uniform) a row with more than 16 entries in the tile, or too far for a buffer offset
        bool b_slow = false;  // this row's entries from `b_from` on go through the guarded loads
        int b_from = 0;
        int b_nv[NVEC];
        u32x4 b_idx[NVEC];
        T b_val[NVEC][4];
        const auto fetch_a = [&](unsigned k) {
            const int64_t i = (int64_t)k * ROWS + lane;
            a_rp = 0;
            a_lo = a_hi = 0;
            if (k < n_rounds && i < n_sel && lane < ROWS) {
                const int64_t row = LIST ? sel[i] : i;
                a_lo = tab[q_tab_index(i, tile, n_tiles)];
                a_hi = tab[q_tab_index(i, tile + 1, n_tiles)];
                a_rp = indptr[row];
            }
        };
        // stage B of the NEXT round, split so that its loads fly behind the current round's work (round 6; the wavefront
        // used to issue them after the round's publication and to need them at once: every round paid one exposed
        // memory latency -- 3 us per wavefront and round whatever the density, 1.6 of the 3.15 ms at config 4):
        //   next_addr  (after step 1 of the current round: its index registers are dead) addresses + INDEX loads,
        //   next_vals  (after step 5: its value registers are dead) VALUE loads, and the next round becomes current.
        constexpr bool EARLY = sizeof(T) == 4 || PIECES == 2;
        int64_t n_base = 0;
        int n_cnt = 0, n_from = 0;
        bool n_more = false, n_slow = false;
        int n_nv[NVEC];
        unsigned n_off[NVEC];
        __amdgpu_buffer_rsrc_t n_vrs = make_rsrc(vals, 0u);
        const auto next_addr = [&]() {
            n_base = a_rp + (int64_t)a_lo;
            n_cnt = (int)(a_hi - a_lo);
            // range-checked 16-byte loads relative to the round's first entry (rows ascend, so it is the first row
            // with entries): lanes without entries and reads past the end of the arrays return zeros
            const unsigned long long has = __builtin_amdgcn_ballot_w64(n_cnt > 0);
            const int first = has ? (int)__builtin_ctzll(has) : 0;
            int64_t e_first = ((int64_t)__builtin_amdgcn_readlane((int)(n_base >> 32), first) << 32) |
                              (unsigned)__builtin_amdgcn_readlane((int)n_base, first);
            if (!has) e_first = e_end;
            const int64_t left = e_end - e_first;
            const unsigned rec = (unsigned)(left < (int64_t)far_limit + SLOTS ? left : (int64_t)far_limit + SLOTS);
            const __amdgpu_buffer_rsrc_t i_rs = make_rsrc(indices + e_first, rec * 4u);
            n_vrs = make_rsrc(vals + e_first, rec * (unsigned)sizeof(T));
            const int64_t rel64 = n_base - e_first;
            const bool far = n_cnt > 0 && rel64 >= (int64_t)far_limit;
            n_slow = far || n_cnt > SLOTS;
            n_from = far ? 0 : SLOTS;
            n_more = __builtin_amdgcn_ballot_w64(n_slow) != 0ull;
            const int rel = far || n_cnt <= 0 ? -1 : (int)rel64;
#pragma unroll
            for (int v = 0; v < NVEC; ++v) {
                const int src = RPV * v + sub;
                const int cnt_v = __shfl(n_cnt, src);
                const int rel_v = __shfl(rel, src);
                int nv = cnt_v - 4 * piece;
                nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
                if (rel_v < 0) nv = 0;
                n_nv[v] = nv;
                n_off[v] = nv > 0 ? (unsigned)(rel_v + 4 * piece) : 0x3fffffffu;  // (no entries: out of range)
                b_idx[v] = __builtin_amdgcn_raw_buffer_load_b128(i_rs, n_off[v] * 4u, 0, 0);
            }
        };
        const auto next_vals = [&]() {
#pragma unroll
            for (int v = 0; v < NVEC; ++v) {
                if constexpr (sizeof(T) == 4) {
                    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(n_vrs, n_off[v] * 4u, 0, 0);
                    b_val[v][0] = __uint_as_float(w.x), b_val[v][1] = __uint_as_float(w.y);
                    b_val[v][2] = __uint_as_float(w.z), b_val[v][3] = __uint_as_float(w.w);
                } else {
                    const unsigned offb = n_nv[v] > 0 ? n_off[v] * 8u : 0xfffffff0u;
                    const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(n_vrs, offb, 0, 0);
                    const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(n_vrs, offb, 16, 0);
                    b_val[v][0] = __hiloint2double((int)w0.y, (int)w0.x);
                    b_val[v][1] = __hiloint2double((int)w0.w, (int)w0.z);
                    b_val[v][2] = __hiloint2double((int)w1.y, (int)w1.x);
                    b_val[v][3] = __hiloint2double((int)w1.w, (int)w1.z);
                }
                b_nv[v] = n_nv[v];
            }
            b_base = n_base;
            b_cnt = n_cnt;
            b_from = n_from;
            b_slow = n_slow;
            b_more = n_more;
        };
        // round k: row masks -> LDS rows needed -> allocation in round order -> zero + scatter -> publication in round
        // order.  The per-entry code has NO branches: a lane without an entry ORs into the trash word behind the masks
        // and scatters into the trash bytes behind them (exec-mask juggling around 32 atomics / stores cost as much as
        // the work itself in the first version).
        // byte offset in LDS of this lane's trash bytes (the scatter of lanes without an entry; behind the ring)
        const unsigned trash_off = (unsigned)(lds_bytes - kQCtl - kChLoaders * kQCmBytes - kQTrash) + 8u * (unsigned)lane;
        const auto write_round = [&](unsigned k) {
            // 1. row masks of the tile's columns
            unsigned c8[NVEC][4];  // 8 x (column inside the tile); 1024 + 8 x lane = the lane's trash word
            const unsigned my_trash = 1024u + 8u * (unsigned)lane;
#pragma unroll
            for (int v = 0; v < NVEC; ++v) {
                const unsigned long long bit = 1ull << (RPV * v + sub);
                const int idx[4] = {(int)b_idx[v].x, (int)b_idx[v].y, (int)b_idx[v].z, (int)b_idx[v].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c8[v][j] = j < b_nv[v] ? (unsigned)(idx[j] - c0) * 8u : my_trash;
                    __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(cm) + c8[v][j]),
                                          bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            if (b_more) {  // (uniform, rare) long or far rows: lane = row, guarded loads
                if (b_slow)
                    for (int j = b_from; j < b_cnt; ++j)
                        __hip_atomic_fetch_or(cm + (indices[b_base + j] - c0), 1ull << lane, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            // (the index registers are dead: the next round's addresses and index loads, the table words of the round after it)
            if constexpr (EARLY) {
                next_addr();
                fetch_a(k + 2 * kChLoaders);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // 2. LDS rows of the round = the longest column queue
            int n = __popcll(cm[lane]);
            if (n_tcols > 64) {
                const int n2 = __popcll(cm[64 + lane]);
                n = n2 > n ? n2 : n;
            }
            n = wave_max_nonneg_dpp(n);
            // 3. rows of the stream, in round order.  A round never wraps around the ring: if its rows do not fit
            // before the end, the rest of the ring becomes zero rows (adding zeros is exact) and the round starts at 0.
            while (q_load(ctl + 0) != k) ICV_Q_WAIT();
            const unsigned start = __builtin_amdgcn_readfirstlane(ctl[1]);
            const int pos0 = __builtin_amdgcn_readfirstlane((int)ctl[2]);
            const int pad = pos0 + n > n_ring ? n_ring - pos0 : 0;
            const int posd = pad ? 0 : pos0;  // first LDS row of the round's entries
            if (lane == 0) {
                int p = posd + n;
                if (p >= n_ring) p -= n_ring;
                ctl[1] = start + (unsigned)(pad + n);
                ctl[2] = (unsigned)p;
                q_store(ctl + 0, k + 1u);
            }
            // the chain must be done with what these ring rows held a lap ago
            const unsigned need = start + (unsigned)(pad + n) - (unsigned)n_ring;
            while ((int)(q_load(ctl + 5) - need) < 0) ICV_Q_WAIT();
            // 4. zeros: the padding rows up to the end of the ring, the round's rows
            unsigned char* z1 = smem + (size_t)pos0 * row_bytes;
            for (int o = lane * 16; o < pad * row_bytes; o += 64 * 16) *reinterpret_cast<uint4*>(z1 + o) = make_uint4(0, 0, 0, 0);
            unsigned char* z2 = smem + (size_t)posd * row_bytes;
            for (int o = lane * 16; o < n * row_bytes; o += 64 * 16) *reinterpret_cast<uint4*>(z2 + o) = make_uint4(0, 0, 0, 0);
            // 5. every entry into LDS row rank(entry) of the round (LDS executes a wavefront's operations in order: the
            // zeros above are in place)
            const unsigned base_off = (unsigned)posd * (unsigned)row_bytes;
#pragma unroll
            for (int v = 0; v < NVEC; ++v) {
                const unsigned long long below = (1ull << (RPV * v + sub)) - 1ull;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long m =
                        *reinterpret_cast<const unsigned long long*>(reinterpret_cast<unsigned char*>(cm) + c8[v][j]);
                    const unsigned rank = (unsigned)__popcll(m & below);
                    // (24-bit multiply: rank < 64, row_bytes <= 512 -- a full-rate v_mad_u32_u24 where the plain product
                    // became a quarter-rate 64-bit mad)
                    unsigned a = base_off + __umul24(rank, (unsigned)row_bytes) + c8[v][j] / (8u / (unsigned)sizeof(T));
                    a = j < b_nv[v] ? a : trash_off;
                    *reinterpret_cast<T*>(smem + a) = b_val[v][j] * scale;
                }
            }
            if (b_more) {
                if (b_slow) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    for (int j = b_from; j < b_cnt; ++j) {
                        const int c = indices[b_base + j] - c0;
                        const int p = posd + __popcll(cm[c] & below);
                        *reinterpret_cast<T*>(smem + (size_t)p * row_bytes + (size_t)c * sizeof(T)) = vals[b_base + j] * scale;
                    }
                }
            }
            // (the value registers are dead: the next round's value loads; it becomes the current round)
            if constexpr (EARLY) next_vals();
            // 6. masks back to zero for this wavefront's next round
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(cm) + lane * 16) = make_uint4(0, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the rows are complete)
            // 7. publish, in round order
            while (q_load(ctl + 3) != k) ICV_Q_WAIT();
            if (lane == 0) {
                ctl[4] = start + (unsigned)(pad + n);
                q_store(ctl + 3, k + 1u);
            }
            if constexpr (!EARLY) {  // (float64 at 16+ slots per row: the early loads' extra registers would spill)
                next_addr();
                next_vals();
                fetch_a(k + 2 * kChLoaders);
            }
        };
        unsigned mine = (unsigned)wave;  // this wavefront's next round
        fetch_a(mine);
        next_addr();
        next_vals();
        fetch_a(mine + kChLoaders);
        for (; mine < n_rounds; mine += kChLoaders) write_round(mine);
    } else {
        const int col = lane * 8 < row_bytes ? c0 + lane * CPL : n_cols;
        lane_t a;
        if constexpr (sizeof(T) == 4) {
            a.x = col < n_cols ? acc[col] : 0.f;
            a.y = col + 1 < n_cols ? acc[col + 1] : 0.f;
        } else {
            a = col < n_cols ? acc[col] : 0.0;
        }
        const int rl = lane * 8 < row_bytes ? lane : 0;  // idle lanes read lane 0's bytes (never stored)
        if (nl == 1) chain_stream<T, 1>(smem, n_ring, n_rounds, rl, a, ctl);
        else if (nl == 2) chain_stream<T, 2>(smem, n_ring, n_rounds, rl, a, ctl);
        else if (nl == 3) chain_stream<T, 3>(smem, n_ring, n_rounds, rl, a, ctl);
        else chain_stream<T, 4>(smem, n_ring, n_rounds, rl, a, ctl);
        if constexpr (sizeof(T) == 4) {
            if (col < n_cols) acc[col] = a.x;
            if (col + 1 < n_cols) acc[col + 1] = a.y;
        } else {
            if (col < n_cols) acc[col] = a;
        }
    }
}

}  // namespace icv

// The public path's noise threshold + CSR pack (reference tl/_infercnv.py:449-455), streamed:
//   k_thr_mask_ring (+ k_thr_mask_ties)   step 5b as a keep-mask + row counts        (icv_threshold_mask)
//   k_csr_fill_ring                        csr_matrix(x_res) from x_res + mask + indptr (icv_csr_fill_masked)
// DESIGN.md 4.6 has the measurements, profiles/r04_pack_experiments.txt every step on the way.
//
// The per-row kernels k_thr_mask / k_csr_fill_masked (one short-lived workgroup / wavefront per row) read x_res at
// 3 TB/s; every variant that gave a wavefront more loads in flight came out slower.  What does stream at 6.3 TB/s on this
// chip is the structure of k_colchain: ONE persistent 1024-thread workgroup per CU whose loader wavefronts copy rows
// HBM -> LDS with LDS-DMA (no registers, no ds_write, the ring IS the bytes in flight -- and with ~5 us of latency under
// load a CU needs ~100 KB of them) while the other wavefronts work on the rows that have landed, one s_barrier per round.
// Here: rows are packed into the ring at their own length (rounded to 16 bytes), kPmRows = 4 adjacent rows per round,
// every slot but the one being read in flight; four loader wavefronts share the 1 KB pieces of a round; twelve consumer
// wavefronts -- three per row, each a contiguous run of the row's 64-window words -- work from LDS.
//
// Three rules made this faster than the per-row kernels (each was measured the hard way):
//   * a CU's 16 wavefronts sit on 4 SIMDs that issue one vector and one scalar instruction per ~4 cycles: ~10
//     instructions per 64 windows is the budget at which HBM, not the consumers, sets the pace;
//   * nothing in a consumer's loop may wait on vmcnt (gfx9 counts a wavefront's loads AND stores in one in-order
//     counter: waiting for a load after a store waits for the store's acknowledgement) -- everything a consumer reads
//     comes through LDS or the scalar cache;
//   * few, wide stores: results are staged in LDS and written by full wavefronts (a dozen short stores per round queue
//     behind the loaders' LDS-DMA instructions in the CU's one vector-memory pipeline).
#pragma once
#include <type_traits>

#include "icv_kernel_chain.hpp"

namespace icv {

constexpr int kPmThreads = 1024;
#ifndef ICV_PM_LOADERS
#define ICV_PM_LOADERS 4
#endif
constexpr int kPmLoaders = ICV_PM_LOADERS;                     // loader wavefronts
constexpr int kPmConsumers = kPmThreads / 64 - kPmLoaders;     // 12
#ifndef ICV_PM_ROWS
#define ICV_PM_ROWS 4
#endif
constexpr int kPmRows = ICV_PM_ROWS;                          // rows per round
constexpr int kPmPerRow = kPmConsumers / kPmRows;              // consumer wavefronts per row
constexpr int kPmMaxSlots = 16;
constexpr int kPmMaxPer = 8;                                   // LDS-DMA loads per loader and round, at most
constexpr int kPmLds = 160 * 1024;

// wait until at most n LDS-DMA loads of this wavefront are outstanding (n wave-uniform, < 64)
__device__ __forceinline__ void pm_wait_vmcnt(int n) {
    switch (n) {
#define ICV_PM_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        ICV_PM_W(1) ICV_PM_W(2) ICV_PM_W(3) ICV_PM_W(4) ICV_PM_W(5) ICV_PM_W(6) ICV_PM_W(7)
        ICV_PM_W(8) ICV_PM_W(9) ICV_PM_W(10) ICV_PM_W(11) ICV_PM_W(12) ICV_PM_W(13) ICV_PM_W(14) ICV_PM_W(15)
        ICV_PM_W(16) ICV_PM_W(17) ICV_PM_W(18) ICV_PM_W(19) ICV_PM_W(20) ICV_PM_W(21) ICV_PM_W(22) ICV_PM_W(23)
        ICV_PM_W(24) ICV_PM_W(25) ICV_PM_W(26) ICV_PM_W(27) ICV_PM_W(28) ICV_PM_W(29) ICV_PM_W(30) ICV_PM_W(31)
        ICV_PM_W(32) ICV_PM_W(33) ICV_PM_W(34) ICV_PM_W(35) ICV_PM_W(36) ICV_PM_W(37) ICV_PM_W(38) ICV_PM_W(39)
        ICV_PM_W(40) ICV_PM_W(41) ICV_PM_W(42) ICV_PM_W(43) ICV_PM_W(44) ICV_PM_W(45) ICV_PM_W(46) ICV_PM_W(47)
        ICV_PM_W(48) ICV_PM_W(49) ICV_PM_W(50) ICV_PM_W(51) ICV_PM_W(52) ICV_PM_W(53) ICV_PM_W(54) ICV_PM_W(55)
        ICV_PM_W(56) ICV_PM_W(57) ICV_PM_W(58) ICV_PM_W(59) ICV_PM_W(60)
#undef ICV_PM_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {  // a wave-uniform value into scalar registers
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}
// a wave-uniform double through the scalar cache (the consumers' threshold: see the second rule above)
__device__ __forceinline__ double scalar_load_f64(const double* p) {
    p = reinterpret_cast<const double*>(uniform_i64(reinterpret_cast<int64_t>(p)));
    double v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// Geometry of the ring for rows of W float32 (host and device agree through this struct)
struct PackRing {
    int row_bytes, row_stride, n_ld, per_loader, n_slots;
    int stage_block, stage_cnt_off;  // per parity: the round's mask words, then its kept counts (one per row and part)
    bool ok;
    __host__ __device__ explicit PackRing(int W) {
        const int n_words = (W + 63) / 64;
        stage_cnt_off = (kPmRows * n_words * 8 + 15) / 16 * 16;
        stage_block = stage_cnt_off + (kPmRows * kPmPerRow * 4 + 15) / 16 * 16;
        row_bytes = W * 4;
        row_stride = (row_bytes + 15) / 16 * 16;
        n_ld = (row_bytes + 1023) / 1024;                                       // 1 KB pieces per row
        per_loader = (kPmRows * n_ld + kPmLoaders - 1) / kPmLoaders;            // LDS-DMA loads per loader and round
        // the last piece of a slot's last row may write up to 1 KB - 16 past the row: lanes past the row are masked off,
        // so the ring needs no slack
        int s = (kPmLds - 2 * stage_block) / (kPmRows * row_stride);
        if (s > kPmMaxSlots) s = kPmMaxSlots;
        while (s > 2 && (s - 1) * per_loader > 60) --s;
        n_slots = s;
        ok = s >= 2 && (s - 1) * per_loader <= 60 && per_loader <= kPmMaxPer && row_bytes >= 1024 && n_words <= 64 * kPmPerRow;
    }
    __host__ __device__ int lds_bytes() const { return n_slots * kPmRows * row_stride + 2 * stage_block;
    }
};

// LDS-DMA with a scalar base: 16 bytes per lane from base + off (off per lane, bytes) to lds_base + 16 * lane
__device__ __forceinline__ void lds_dma16_s(const void* base, unsigned off, unsigned lds_base) {
    // (the compiler keeps some wave-uniform values in vector registers and does not legalise asm operands)
    base = reinterpret_cast<const void*>(uniform_i64(reinterpret_cast<int64_t>(base)));
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(base), "s"(lds_base)
        : "memory");
}

// One mask word of the consumer loop below: y = the window's bits, lo / span the threshold as patterns (see there).
// The lane masks come straight from the compares; word T of the wavefront goes to lane T of (mine_lo, mine_hi).
#define ICV_PM_WORD(T, Y)                                                                                              \
    {                                                                                                                  \
        const unsigned d_ = ((Y) & 0x7fffffffu) - lo;                                                                  \
        const unsigned long long m_ = __builtin_amdgcn_sicmp((int)d_, 0, 39 /* >= */);                                 \
        tie_any |= __builtin_amdgcn_uicmp(d_, span, 37 /* <= */);                                                      \
        kept += __builtin_popcountll(m_);                                                                              \
        asm volatile("v_writelane_b32 %0, %1, " #T : "+v"(mine_lo) : "s"((unsigned)m_));                               \
        asm volatile("v_writelane_b32 %0, %1, " #T : "+v"(mine_hi) : "s"((unsigned)(m_ >> 32)));                       \
    }

// mask: n_rows x n_words uint64 (bit j & 63 of word j >> 6 = window j kept), row_nnz: kept windows of the row.
// tie_n (zero on entry) / tie_list (n_rows * kPmPerRow entries): the (row, wavefront part) pairs with undecided windows.
// grid = CUs (fewer rounds: fewer workgroups); dynamic LDS = PackRing::lds_bytes().
// Rows of `out` start on 16-byte boundaries (the caller checks; else the per-row kernel k_thr_mask runs).
__global__ void __launch_bounds__(kPmThreads) k_thr_mask_ring(const KParams P, const double* thr, int64_t chunksize,
                                                              int64_t row_phase, unsigned long long* mask, int n_words,
                                                              int64_t* row_nnz, unsigned* tie_n, unsigned long long* tie_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const PackRing g(P.W);
    const int64_t total_rounds = (P.n_rows + kPmRows - 1) / kPmRows;
    const int64_t n_mine = total_rounds > blockIdx.x ? (total_rounds - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int slot_bytes = kPmRows * g.row_stride;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const bool has_thr = thr != nullptr;
    const int64_t stride_rows = (int64_t)gridDim.x * kPmRows;  // rows between two rounds of this workgroup
    // A SIMD issues one vector AND one scalar instruction per four cycles, shared by the four wavefronts it hosts here:
    // everything per round is kept incremental (no division, no 64-bit multiply), in both roles.

    if (wave < kPmLoaders) {
        // ---- loaders: the round's kPmRows * n_ld pieces dealt round robin; every loader issues per_loader loads a round
        // (a loader short of pieces repeats piece 0 of row 0: same bytes to the same place).  The piece table does not
        // change from round to round.
        unsigned p_dst[kPmMaxPer], p_off[kPmMaxPer];
        int p_row[kPmMaxPer];
        bool p_on[kPmMaxPer];
#pragma unroll
        for (int t = 0; t < kPmMaxPer; ++t) {
            int q = t * kPmLoaders + wave;
            if (q >= kPmRows * g.n_ld) q = 0;
            const int r = q / g.n_ld, piece = q - r * g.n_ld;
            p_row[t] = r;
            p_dst[t] = (unsigned)r * (unsigned)g.row_stride + (unsigned)piece * 1024u;
            p_off[t] = (unsigned)(piece * 64 + lane) * 16u;
            // the piece that holds the row's last windows may run past W: it stays inside the row's ldo
            p_on[t] = t < g.per_loader && (int)p_off[t] < g.row_bytes;
        }
        const int64_t row_step_bytes = P.ldo * 4;
        const unsigned char* issue_base = reinterpret_cast<const unsigned char*>(P.out) + (int64_t)blockIdx.x * kPmRows * row_step_bytes;
        int64_t issue_row0 = (int64_t)blockIdx.x * kPmRows;
        int issue_slot = 0;
        const auto issue = [&]() {
            const unsigned slot_lds = lds0 + (unsigned)issue_slot * (unsigned)slot_bytes;
            const unsigned char* rb[kPmRows];
#pragma unroll
            for (int r = 0; r < kPmRows; ++r)  // (past the last row: row 0 of the round again, never decided)
                rb[r] = issue_row0 + r < P.n_rows ? issue_base + r * row_step_bytes : issue_base;
#pragma unroll
            for (int t = 0; t < kPmMaxPer; ++t) {
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_PM_EXP_NOLOAD)
                continue;
#endif
                if (t < g.per_loader) {  // (uniform)
                    const unsigned char* base = rb[0];
#pragma unroll
                    for (int r = 1; r < kPmRows; ++r) base = p_row[t] == r ? rb[r] : base;
                    if (p_on[t]) lds_dma16_s(base, p_off[t], slot_lds + p_dst[t]);
                }
            }
            issue_slot = issue_slot + 1 == g.n_slots ? 0 : issue_slot + 1;
            issue_base += stride_rows * row_step_bytes;
            issue_row0 += stride_rows;
        };
        const int D = g.n_slots - 1;  // rounds in flight
        for (int64_t k = 0; k < D - 1 && k < n_mine; ++k) issue();
        for (int64_t k = 0; k < n_mine; ++k) {
            if (k + D - 1 < n_mine) {
                issue();
                pm_wait_vmcnt((D - 1) * g.per_loader);  // round k has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();  // (the consumers' last hand-over, see there)
    } else {
        // ---- consumers: wavefront c works on row c / kPmPerRow of the round, a contiguous run of its mask words -----
        // The decision is made on the bit patterns (|y| and the threshold are non-negative floats: their patterns are
        // ordered): with a = |y|'s pattern and b the threshold's, thr_compare's "kept" (a != 0, a - b >= -1) is
        // a >= lo = max(b - 1, 1), and "float32 cannot decide" (|a - b| <= 1) is a - lo <= span = b + 1 - lo.  Three
        // vector instructions a word; ties only raise a flag here.  A NaN or negative threshold keeps every stored
        // value: b = 0 does that.
        const int c = wave - kPmLoaders, r_in = c / kPmPerRow, part = c - r_in * kPmPerRow;
        const int wpw = (n_words + kPmPerRow - 1) / kPmPerRow;  // words per wavefront (<= 64)
        const int u_begin = part * wpw;
        const int n_u = u_begin >= n_words ? 0 : (n_words - u_begin < wpw ? n_words - u_begin : wpw);
        const bool owns_last = n_u > 0 && u_begin + n_u == n_words;
        const unsigned long long tail_mask = (P.W & 63) ? (1ull << (P.W & 63)) - 1ull : ~0ull;  // of the row's last word
        const unsigned lane_off = (unsigned)(u_begin * 64 + lane) * 4u;
        int64_t row = (int64_t)blockIdx.x * kPmRows + r_in;
        // chunk of the row (its threshold), kept incrementally
        int64_t ch = 0, rem = 0, ch_step = 0, rem_step = 0;
        if (has_thr) {
            ch = uniform_i64((row + row_phase) / chunksize);  // (the 64-bit division is vector code)
            rem = (row + row_phase) - ch * chunksize;
            ch_step = uniform_i64(stride_rows / chunksize);
            rem_step = stride_rows - ch_step * chunksize;
        }
        // The results leave through LDS: every wavefront puts its words and its kept count into the round's staging
        // block (two blocks, alternating), and after the next barrier ONE wavefront writes the round -- kPmRows adjacent
        // mask rows, contiguous in memory -- with a couple of full-width stores.  Twelve short stores and twelve atomics a
        // round queue behind the loaders' LDS-DMA instructions in the CU's one vector-memory pipeline and stall the
        // wavefronts that issue them: measured 0.05 ms of 0.2.
        unsigned char* const stage0 = smem + (size_t)g.n_slots * slot_bytes;
        const unsigned my_word_off = (unsigned)(r_in * n_words + u_begin + lane) * 8u;
        const unsigned my_cnt_off = (unsigned)g.stage_cnt_off + (unsigned)(r_in * kPmPerRow + part) * 4u;
        const bool storer = c == kPmConsumers - 1;
        const auto flush = [&](int64_t k) {  // round k of this workgroup: staging block k & 1 -> mask, row_nnz
            const unsigned char* st = stage0 + (size_t)(k & 1) * g.stage_block;
            const int64_t row0 = ((int64_t)blockIdx.x + k * gridDim.x) * kPmRows;
            const int valid = P.n_rows - row0 < kPmRows ? (int)(P.n_rows - row0) : kPmRows;
            unsigned long long* dst = mask + row0 * (int64_t)n_words;
            for (int w = lane; w < valid * n_words; w += 64) dst[w] = *reinterpret_cast<const unsigned long long*>(st + w * 8);
            if (lane < valid) {
                int tot = 0;
#pragma unroll
                for (int q = 0; q < kPmPerRow; ++q) tot += *reinterpret_cast<const int*>(st + g.stage_cnt_off + (lane * kPmPerRow + q) * 4);
                row_nnz[row0 + lane] = tot;
            }
        };
        // the loop over the rounds, compiled once per word count 1..16 (straight-line code: a jump per word costs as much
        // as the word) and once for longer runs (NW = 17: sweeps of eight words)
        const auto consume = [&](auto nw_tag) {
            constexpr int NW = decltype(nw_tag)::value;
            int slot_i = 0;
            double th = has_thr && row < P.n_rows ? scalar_load_f64(thr + ch) : 0.0;
            for (int64_t k = 0; k < n_mine; ++k) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");  // the slot was written by the loaders' DMA: read it after the barrier
                const unsigned char* lrow = smem + (size_t)slot_i * slot_bytes + (size_t)r_in * g.row_stride + lane_off;
                slot_i = slot_i + 1 == g.n_slots ? 0 : slot_i + 1;
                unsigned char* stage = stage0 + (size_t)(k & 1) * g.stage_block;
                if (storer && k > 0) flush(k - 1);
                if (lane == 0) *reinterpret_cast<int*>(stage + my_cnt_off) = 0;
                bool skip = row >= P.n_rows || NW == 0;  // (uniform; the barrier count is kept by the loop header)
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_PM_EXP_NOCONSUME)
                skip = true;
#endif
                if (!skip) {
                    const float thf = (float)th;
                    const int thb = thf > 0.0f ? __float_as_int(thf) : 0;
                    const unsigned lo = thb > 2 ? (unsigned)thb - 1u : 1u, span = (unsigned)thb + 1u - lo;
                    unsigned long long mine = 0, tie_any = 0;
                    int kept = 0;
                    // (a read past the row's end -- the last word, in the sweeps the words of the next wavefront --
                    // returns other rows' bytes or, past the allocation, zero: those windows are masked / never used)
#define ICV_PM_Y(I) (*reinterpret_cast<const unsigned*>(lw + (I) * 256))
                    if constexpr (NW <= 16) {
                        const unsigned char* lw = lrow;
                        unsigned y[16];
#pragma unroll
                        for (int t = 0; t < 16; ++t) y[t] = t < NW ? ICV_PM_Y(t) : 0u;
                        int mine_lo = 0, mine_hi = 0;
                        if constexpr (NW > 15) ICV_PM_WORD(15, y[15])
                        if constexpr (NW > 14) ICV_PM_WORD(14, y[14])
                        if constexpr (NW > 13) ICV_PM_WORD(13, y[13])
                        if constexpr (NW > 12) ICV_PM_WORD(12, y[12])
                        if constexpr (NW > 11) ICV_PM_WORD(11, y[11])
                        if constexpr (NW > 10) ICV_PM_WORD(10, y[10])
                        if constexpr (NW > 9) ICV_PM_WORD(9, y[9])
                        if constexpr (NW > 8) ICV_PM_WORD(8, y[8])
                        if constexpr (NW > 7) ICV_PM_WORD(7, y[7])
                        if constexpr (NW > 6) ICV_PM_WORD(6, y[6])
                        if constexpr (NW > 5) ICV_PM_WORD(5, y[5])
                        if constexpr (NW > 4) ICV_PM_WORD(4, y[4])
                        if constexpr (NW > 3) ICV_PM_WORD(3, y[3])
                        if constexpr (NW > 2) ICV_PM_WORD(2, y[2])
                        if constexpr (NW > 1) ICV_PM_WORD(1, y[1])
                        if constexpr (NW > 0) ICV_PM_WORD(0, y[0])
                        mine = ((unsigned long long)(unsigned)mine_hi << 32) | (unsigned)mine_lo;
                    } else {
                        for (int i0 = 0; i0 < n_u; i0 += 8) {
                            const unsigned char* lw = lrow + i0 * 256;
                            const unsigned y0 = ICV_PM_Y(0), y1 = ICV_PM_Y(1), y2 = ICV_PM_Y(2), y3 = ICV_PM_Y(3), y4 = ICV_PM_Y(4),
                                           y5 = ICV_PM_Y(5), y6 = ICV_PM_Y(6), y7 = ICV_PM_Y(7);
                            int mine_lo = 0, mine_hi = 0;
                            switch (n_u - i0) {  // (uniform) the words there are, last first
                                default: ICV_PM_WORD(7, y7)
                                case 7: ICV_PM_WORD(6, y6)
                                case 6: ICV_PM_WORD(5, y5)
                                case 5: ICV_PM_WORD(4, y4)
                                case 4: ICV_PM_WORD(3, y3)
                                case 3: ICV_PM_WORD(2, y2)
                                case 2: ICV_PM_WORD(1, y1)
                                case 1: ICV_PM_WORD(0, y0)
                            }
                            // lanes 0..7 of this sweep are words i0..i0 + 7
                            const unsigned long long w8 = ((unsigned long long)(unsigned)mine_hi << 32) | (unsigned)mine_lo;
                            const unsigned long long sh = __shfl(w8, (lane - i0) & 63, 64);
                            if (lane >= i0 && lane < i0 + 8) mine = sh;
                        }
                    }
#undef ICV_PM_Y
                    if (owns_last) {  // windows past W in the row's last word
                        const unsigned long long last = __shfl(mine, n_u - 1, 64);
                        kept -= __builtin_popcountll(last & ~tail_mask);
                        if (lane == n_u - 1) mine &= tail_mask;
                    }
                    // ties (about one window in 1e7; none without a threshold): every flagged window is recomputed in
                    // float64 from the input, each by its own lane
                    // ties (about one window in 1e7; none without a threshold): the wavefront's words of this row go on
                    // a list, k_thr_mask_ties recomputes the flagged windows in float64 after this kernel.  (The code that
                    // does it is large; a call from here, even one never taken, costs the loop a quarter of its speed.)
                    if (has_thr && tie_any != 0ull && lane == 0) tie_list[atomicAdd(tie_n, 1u)] = ((unsigned long long)row << 8) | (unsigned)part;
                    if (lane < n_u) *reinterpret_cast<unsigned long long*>(stage + my_word_off) = mine;
                    if (lane == 0) *reinterpret_cast<int*>(stage + my_cnt_off) = kept;
                }
                // the next round's row and threshold
                row += stride_rows;
                if (has_thr) {
                    const int64_t ch_was = ch;
                    ch += ch_step;
                    rem += rem_step;
                    if (rem >= chunksize) {
                        rem -= chunksize;
                        ++ch;
                    }
                    if (ch != ch_was && row < P.n_rows) th = scalar_load_f64(thr + ch);
                }
            }
            __builtin_amdgcn_s_barrier();  // the last round is staged
            if (storer && n_mine > 0) flush(n_mine - 1);
        };
        switch (n_u) {  // (uniform)
            case 0: consume(std::integral_constant<int, 0>{}); break;
            case 1: consume(std::integral_constant<int, 1>{}); break;
            case 2: consume(std::integral_constant<int, 2>{}); break;
            case 3: consume(std::integral_constant<int, 3>{}); break;
            case 4: consume(std::integral_constant<int, 4>{}); break;
            case 5: consume(std::integral_constant<int, 5>{}); break;
            case 6: consume(std::integral_constant<int, 6>{}); break;
            case 7: consume(std::integral_constant<int, 7>{}); break;
            case 8: consume(std::integral_constant<int, 8>{}); break;
            case 9: consume(std::integral_constant<int, 9>{}); break;
            case 10: consume(std::integral_constant<int, 10>{}); break;
            case 11: consume(std::integral_constant<int, 11>{}); break;
            case 12: consume(std::integral_constant<int, 12>{}); break;
            case 13: consume(std::integral_constant<int, 13>{}); break;
            case 14: consume(std::integral_constant<int, 14>{}); break;
            case 15: consume(std::integral_constant<int, 15>{}); break;
            case 16: consume(std::integral_constant<int, 16>{}); break;
            default: consume(std::integral_constant<int, 17>{}); break;
        }
    }
}
#undef ICV_PM_WORD

// The flagged windows of k_thr_mask_ring's list, recomputed in float64 from the input (canonical order); bits cleared in
// the mask, row counts lowered.  One workgroup per list entry (a row and one wavefront's run of its words): the threads
// find the flagged windows, then -- as k_thr_mask does -- stage each window's genes in LDS together and thread 0 adds them
// in the canonical order.  The last workgroup to finish zeroes the two counters for the next call (tie_n[0] = entries,
// tie_n[1] = workgroups done).
template <typename T, bool CSR>
__global__ void __launch_bounds__(256) k_thr_mask_ties(const KParams P, const double* thr, int64_t chunksize, int64_t row_phase,
                                                       unsigned long long* mask, int n_words, int64_t* row_nnz, unsigned* tie_n,
                                                       const unsigned long long* tie_list) {
    __shared__ int n_t, n_dropped;
    __shared__ int tj[64];
    __shared__ double vals[kTieBuf];
    const unsigned n = __hip_atomic_load(tie_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int wpw = (n_words + kPmPerRow - 1) / kPmPerRow;
    for (unsigned e = blockIdx.x; e < n; e += gridDim.x) {
        const unsigned long long item = tie_list[e];
        const int64_t row = (int64_t)(item >> 8);
        const int part = (int)(item & 255), u_begin = part * wpw;
        const int n_u = n_words - u_begin < wpw ? n_words - u_begin : wpw;
        const double th = thr[(row + row_phase) / chunksize];
        const float thf = (float)th;
        const int thb = thf > 0.0f ? __float_as_int(thf) : 0;
        const unsigned lo = thb > 2 ? (unsigned)thb - 1u : 1u, span = (unsigned)thb + 1u - lo;
        unsigned long long* mrow = mask + row * (int64_t)n_words;
        const auto drop_window = [&](int j) {
            atomicAnd(mrow + (j >> 6), ~(1ull << (j & 63)));
            atomicAdd(&n_dropped, 1);
        };
        if (threadIdx.x == 0) {
            n_t = 0;
            n_dropped = 0;
        }
        __syncthreads();
        for (int w = threadIdx.x; w < n_u * 64; w += 256) {
            const int j = u_begin * 64 + w;
            if (j >= P.W) continue;
            const unsigned d = (__float_as_uint(P.out[row * P.ldo + j]) & 0x7fffffffu) - lo;
            if ((int)d < 0 || d > span) continue;
            const int idx = atomicAdd(&n_t, 1);
            if (idx < 64) {
                tj[idx] = j;
            } else {  // more than 64 undecided windows in one run of words (a row equal to the reference): serially
                const int st = P.w_start[j];
                const double yd = window_canonical(P, j, [&](int kk) { return value_at<T, CSR>(P, row, st + kk); }) - P.cell_median[row];
                if (fabs(yd) < th) drop_window(j);
            }
        }
        __syncthreads();
        const int nt = n_t < 64 ? n_t : 64;
        for (int i = 0; i < nt; ++i) {
            const int j = tj[i];
            const int st = P.w_start[j], ln = P.w_len[j];
            const int len = ln > 0 ? ln : -ln;
            if (len <= kTieBuf) {
                for (int k = threadIdx.x; k < len; k += 256) vals[k] = value_at<T, CSR>(P, row, st + k);
                __syncthreads();
                if (threadIdx.x == 0 && fabs(window_canonical(P, j, [&](int k) { return vals[k]; }) - P.cell_median[row]) < th) drop_window(j);
                __syncthreads();
            } else if (threadIdx.x == 0) {
                if (fabs(window_canonical(P, j, [&](int k) { return value_at<T, CSR>(P, row, st + k); }) - P.cell_median[row]) < th) drop_window(j);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && n_dropped)
            atomicAdd(reinterpret_cast<unsigned long long*>(row_nnz + row), (unsigned long long)(-(long long)n_dropped));
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(tie_n + 1, 1u) == gridDim.x - 1) {
            tie_n[0] = 0;
            tie_n[1] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// csr_matrix(x_res) from the keep-mask, streamed: k_csr_fill_ring.
//
// k_csr_fill_masked (one wavefront per row) issues a load and two scattered stores per 64 windows: 87 vector-memory
// instructions per row of config 2, and the CU's one vector-memory pipeline takes ~16 cycles for each -- 0.24 ms per
// 100 000 cells whatever the memory system could do.  Here the rows come through the LDS ring of k_thr_mask_ring (the
// round's mask rows and row offsets ride along as extra LDS-DMA pieces of loader 0), the consumers compact the kept
// windows of the round -- kPmRows rows, adjacent in the output -- into an LDS staging area as {column, value bits}
// pairs, and in the NEXT round (after its barrier; two staging blocks alternate) all of them copy the block out with
// full-width stores (column as int32, value widened to float64 on the way): a few dozen vector-memory instructions per
// round instead of 350, one barrier per round, the loaders never wait for the consumers' phases.  A round with more kept
// windows than a staging block holds (no threshold, dense rows) is done in passes with two more barriers each.
struct FillRing {
    int row_bytes, row_stride, n_ld, per_loader, n_slots;
    int n_words, mask_off, n_mask_ld, ip_off, slot_total, n_extra;
    int stage_off, cap;  // staging: byte offset in LDS, entries (8 bytes each) per block
    bool ok;
    __host__ __device__ explicit FillRing(int W) {
        n_words = (W + 63) / 64;
        row_bytes = W * 4;
        row_stride = (row_bytes + 15) / 16 * 16;
        n_ld = (row_bytes + 1023) / 1024;
        per_loader = (kPmRows * n_ld + kPmLoaders - 1) / kPmLoaders;
        mask_off = kPmRows * row_stride;
        n_mask_ld = (kPmRows * n_words * 8 + 255) / 256;  // 256-byte pieces (4 bytes per lane: exact to the word)
        ip_off = mask_off + n_mask_ld * 256;
        slot_total = ip_off + 256;                         // kPmRows + 1 row offsets in one piece
        n_extra = n_mask_ld + 1;
        int s = (kPmLds - 16 * 1024) / slot_total;         // at least 2 048 staging entries
        if (s > kPmMaxSlots) s = kPmMaxSlots;
        while (s > 2 && (s - 1) * (per_loader + n_extra) > 60) --s;
        n_slots = s;
        stage_off = s * slot_total;
        cap = s >= 2 ? (kPmLds - stage_off) / 2 / 8 / 64 * 64 : 0;  // two blocks, alternating
        ok = s >= 2 && (s - 1) * (per_loader + n_extra) <= 60 && per_loader <= kPmMaxPer && row_bytes >= 1024 &&
             n_words <= 64 && cap >= 1024;
    }
    __host__ __device__ int lds_bytes() const { return stage_off + 2 * cap * 8; }
};

// LDS-DMA, 4 bytes per lane: from base + off (off per lane, bytes) to lds_base + 4 * lane
__device__ __forceinline__ void lds_dma4_s(const void* base, unsigned off, unsigned lds_base) {
    base = reinterpret_cast<const void*>(uniform_i64(reinterpret_cast<int64_t>(base)));
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(base), "s"(lds_base)
                 : "memory");
}
__device__ __forceinline__ int64_t lds_uniform_i64(const unsigned char* p) {  // a wave-uniform LDS word into scalar registers
    return uniform_i64(*reinterpret_cast<const int64_t*>(p));
}

// x: n_rows x n_cols float32 (rows 16-byte aligned), mask / indptr as written by the mask pass and icv_row_offsets;
// indices / data: the packed output.  grid = CUs; dynamic LDS = FillRing::lds_bytes().
__global__ void __launch_bounds__(kPmThreads) k_csr_fill_ring(const float* x, int64_t n_rows, int n_cols, int64_t ld,
                                                              const unsigned long long* mask, const int64_t* indptr,
                                                              int32_t* indices, double* data) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const FillRing g(n_cols);
    const int n_words = g.n_words;
    const int64_t total_rounds = (n_rows + kPmRows - 1) / kPmRows;
    const int64_t n_mine = total_rounds > blockIdx.x ? (total_rounds - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const int64_t stride_rows = (int64_t)gridDim.x * kPmRows;
    // passes over the staging area for a round with n kept windows (every wavefront computes it from the round's row
    // offsets in LDS: the barriers of a round are 2 * passes, for loaders and consumers alike)
    const auto passes_of = [&](int n) { return n > g.cap ? (n + g.cap - 1) / g.cap : 1; };

    if (wave < kPmLoaders) {
        unsigned p_dst[kPmMaxPer], p_off[kPmMaxPer];
        int p_row[kPmMaxPer];
        bool p_on[kPmMaxPer];
#pragma unroll
        for (int t = 0; t < kPmMaxPer; ++t) {
            int q = t * kPmLoaders + wave;
            if (q >= kPmRows * g.n_ld) q = 0;
            const int r = q / g.n_ld, piece = q - r * g.n_ld;
            p_row[t] = r;
            p_dst[t] = (unsigned)r * (unsigned)g.row_stride + (unsigned)piece * 1024u;
            p_off[t] = (unsigned)(piece * 64 + lane) * 16u;
            p_on[t] = t < g.per_loader && (int)p_off[t] < g.row_bytes;
        }
        const int64_t row_step_bytes = ld * 4;
        const unsigned char* issue_base = reinterpret_cast<const unsigned char*>(x) + (int64_t)blockIdx.x * kPmRows * row_step_bytes;
        int64_t issue_row0 = (int64_t)blockIdx.x * kPmRows;
        int issue_slot = 0;
        const int n_extra = wave == 0 ? g.n_extra : 0;
        const auto issue = [&]() {
            const unsigned slot_lds = lds0 + (unsigned)issue_slot * (unsigned)g.slot_total;
            const unsigned char* rb[kPmRows];
#pragma unroll
            for (int r = 0; r < kPmRows; ++r) rb[r] = issue_row0 + r < n_rows ? issue_base + r * row_step_bytes : issue_base;
#pragma unroll
            for (int t = 0; t < kPmMaxPer; ++t) {
                if (t < g.per_loader) {  // (uniform)
                    const unsigned char* base = rb[0];
#pragma unroll
                    for (int r = 1; r < kPmRows; ++r) base = p_row[t] == r ? rb[r] : base;
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_FR_EXP_NOLOAD)
                    (void)base;
#else
                    if (p_on[t]) lds_dma16_s(base, p_off[t], slot_lds + p_dst[t]);
#endif
                }
            }
            if (wave == 0) {
                // the round's mask rows and kPmRows + 1 row offsets.  Every piece is issued with at least one lane (the
                // count of outstanding loads must not depend on the data): lanes past the valid bytes re-read byte 0
                // into their own, unused place
                const int valid = n_rows - issue_row0 < kPmRows ? (int)(n_rows - issue_row0) : kPmRows;
                const unsigned char* mb = reinterpret_cast<const unsigned char*>(mask + issue_row0 * (int64_t)n_words);
                const unsigned m_valid = (unsigned)(valid * n_words * 8);
                for (int pc = 0; pc < g.n_mask_ld; ++pc) {
                    const unsigned off = (unsigned)(pc * 64 + lane) * 4u;
                    lds_dma4_s(mb, off < m_valid ? off : 0u, slot_lds + (unsigned)g.mask_off + (unsigned)pc * 256u);
                }
                const unsigned ioff = (unsigned)lane * 4u;
                lds_dma4_s(indptr + issue_row0, ioff < (unsigned)(valid + 1) * 8u ? ioff : 0u, slot_lds + (unsigned)g.ip_off);
            }
            issue_slot = issue_slot + 1 == g.n_slots ? 0 : issue_slot + 1;
            issue_base += stride_rows * row_step_bytes;
            issue_row0 += stride_rows;
        };
        const int D = g.n_slots - 1;
        int slot_i = 0;
        int64_t row0 = (int64_t)blockIdx.x * kPmRows;
        for (int64_t k = 0; k < D - 1 && k < n_mine; ++k) issue();
        for (int64_t k = 0; k < n_mine; ++k) {
            if (k + D - 1 < n_mine) {
                issue();
                pm_wait_vmcnt((D - 1) * (g.per_loader + n_extra));  // round k has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();  // A
            asm volatile("" ::: "memory");
            const unsigned char* sl = smem + (size_t)slot_i * g.slot_total;
            slot_i = slot_i + 1 == g.n_slots ? 0 : slot_i + 1;
            const int valid = n_rows - row0 < kPmRows ? (int)(n_rows - row0) : kPmRows;
            const int n_round = (int)(lds_uniform_i64(sl + g.ip_off + 8 * valid) - lds_uniform_i64(sl + g.ip_off));
            const int np = passes_of(n_round);
            if (np > 1)
                for (int b = 0; b < 2 * np; ++b) __builtin_amdgcn_s_barrier();  // (staged, copied) per pass
            row0 += stride_rows;
        }
        __builtin_amdgcn_s_barrier();  // the consumers' last block
    } else {
        const int c = wave - kPmLoaders, r_in = c / kPmPerRow, part = c - r_in * kPmPerRow;
        const int wpw = (n_words + kPmPerRow - 1) / kPmPerRow;
        const int u_begin = part * wpw;
        const int n_u = u_begin >= n_words ? 0 : (n_words - u_begin < wpw ? n_words - u_begin : wpw);
        const unsigned lane_off = (unsigned)(u_begin * 64 + lane) * 4u;
        unsigned char* const stage0 = smem + g.stage_off;
        const int j_lane = u_begin * 64 + lane;
        // n entries of a staging block -> indices / data at entry `at` (all consumer wavefronts, 64 entries a step)
        const auto copy_out = [&](const unsigned char* st, int n, int64_t at) {
            int32_t* oi = indices + at;
            double* od = data + at;
#if defined(ICV_DEV_EXPERIMENTS) && (defined(ICV_FR_EXP_NOCONSUME) || defined(ICV_FR_EXP_NOCOPY))
            n = 0;
#endif
            for (int e = c * 64 + lane; e < n; e += kPmConsumers * 64) {
                const uint2 v = *reinterpret_cast<const uint2*>(st + (size_t)e * 8);
                oi[e] = (int32_t)v.x;
                od[e] = (double)__uint_as_float(v.y);
            }
        };
        const auto consume = [&](auto nw_tag) {
            constexpr int NW = decltype(nw_tag)::value;
            int slot_i = 0, pend_n = 0;
            int64_t pend_at = 0;
            int64_t row0 = (int64_t)blockIdx.x * kPmRows;
            for (int64_t k = 0; k < n_mine; ++k) {
                __builtin_amdgcn_s_barrier();  // the round has landed; the previous round's staging block is complete
                asm volatile("" ::: "memory");
                const unsigned char* sl = smem + (size_t)slot_i * g.slot_total;
                slot_i = slot_i + 1 == g.n_slots ? 0 : slot_i + 1;
                const int valid = n_rows - row0 < kPmRows ? (int)(n_rows - row0) : kPmRows;
                const int64_t ip0 = lds_uniform_i64(sl + g.ip_off);
                const int n_round = (int)(lds_uniform_i64(sl + g.ip_off + 8 * valid) - ip0);
#if defined(ICV_DEV_EXPERIMENTS) && defined(ICV_FR_EXP_NOCONSUME)
                const bool have = false;
#else
                const bool have = r_in < valid && NW > 0;
#endif
                // this wavefront's first output position inside the round: the row's offset + the kept windows of the
                // row's earlier words
                unsigned long long wl = 0;
                int base0 = 0;
                if (have) {
                    if (lane < n_words) wl = *reinterpret_cast<const unsigned long long*>(sl + g.mask_off + (r_in * n_words + lane) * 8);
                    const int pc = __builtin_popcountll(wl);
                    const int excl = wave_scan_dpp(pc) - pc;
                    base0 = (int)(lds_uniform_i64(sl + g.ip_off + 8 * r_in) - ip0) + __builtin_amdgcn_readlane(excl, u_begin);
                }
                const unsigned char* lrow = sl + (size_t)r_in * g.row_stride + lane_off;
                const int np = passes_of(n_round);
                unsigned char* const stage = stage0 + (size_t)(k & 1) * g.cap * 8;
                const unsigned stage_lds = lds0 + (unsigned)g.stage_off + (unsigned)(k & 1) * (unsigned)g.cap * 8u;
                // the previous round's block (complete: every wavefront has passed this round's barrier)
                if (pend_n > 0) copy_out(stage0 + (size_t)((k - 1) & 1) * g.cap * 8, pend_n, pend_at);
                pend_n = 0;
                if (np == 1) {
                    // the usual round (everything fits the block): the mask word IS the lane mask of the write -- no test
                    // per lane, no branch, the row's values all requested before the first use
                    if (have) {
                        if constexpr (NW <= 16) {
                            unsigned yb[NW > 0 ? NW : 1];
#pragma unroll
                            for (int t = 0; t < NW; ++t) yb[t] = *reinterpret_cast<const unsigned*>(lrow + t * 256);
                            int b = base0;
#pragma unroll
                            for (int t = 0; t < NW; ++t) {
                                const unsigned m_lo = __builtin_amdgcn_readlane((unsigned)wl, u_begin + t);
                                const unsigned m_hi = __builtin_amdgcn_readlane((unsigned)(wl >> 32), u_begin + t);
                                const int rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0));
                                const unsigned addr = stage_lds + (unsigned)(b + rank) * 8u;
                                const unsigned long long val = ((unsigned long long)yb[t] << 32) | (unsigned)(j_lane + t * 64);
                                const unsigned long long m = ((unsigned long long)m_hi << 32) | m_lo;
                                unsigned long long sv;
                                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0"
                                             : "=&s"(sv)
                                             : "s"(m), "v"(addr), "v"(val)
                                             : "memory");
                                b += __builtin_popcount(m_lo) + __builtin_popcount(m_hi);
                            }
                        } else {
                            int b = base0;
                            for (int t = 0; t < n_u; ++t) {
                                const unsigned long long m = __shfl(wl, u_begin + t, 64);
                                const unsigned yv = *reinterpret_cast<const unsigned*>(lrow + t * 256);
                                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                                if ((m >> lane) & 1ull)
                                    *reinterpret_cast<uint2*>(stage + (size_t)(b + rank) * 8) = make_uint2((unsigned)(j_lane + t * 64), yv);
                                b += __builtin_popcountll(m);
                            }
                        }
                    }
                    pend_n = n_round;  // copied out in the next round
                    pend_at = ip0;
                } else {
                    for (int p = 0; p < np; ++p) {
                        const int p0 = p * g.cap;
                        if (have) {
                            int b = base0 - p0;
                            for (int t = 0; t < n_u; ++t) {
                                const unsigned long long m = __shfl(wl, u_begin + t, 64);
                                const unsigned yv = *reinterpret_cast<const unsigned*>(lrow + t * 256);
                                const unsigned pos = (unsigned)(b + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0)));
                                if (((m >> lane) & 1ull) && pos < (unsigned)g.cap)
                                    *reinterpret_cast<uint2*>(stage + (size_t)pos * 8) = make_uint2((unsigned)(j_lane + t * 64), yv);
                                b += __builtin_popcountll(m);
                            }
                        }
                        __builtin_amdgcn_s_barrier();  // the pass is staged
                        copy_out(stage, n_round - p0 < g.cap ? n_round - p0 : g.cap, ip0 + p0);
                        __builtin_amdgcn_s_barrier();  // ... and copied: the block is free
                    }
                }
                row0 += stride_rows;
            }
            __builtin_amdgcn_s_barrier();
            if (pend_n > 0) copy_out(stage0 + (size_t)((n_mine - 1) & 1) * g.cap * 8, pend_n, pend_at);
        };
        switch (n_u) {  // (uniform)
            case 0: consume(std::integral_constant<int, 0>{}); break;
            case 1: consume(std::integral_constant<int, 1>{}); break;
            case 2: consume(std::integral_constant<int, 2>{}); break;
            case 3: consume(std::integral_constant<int, 3>{}); break;
            case 4: consume(std::integral_constant<int, 4>{}); break;
            case 5: consume(std::integral_constant<int, 5>{}); break;
            case 6: consume(std::integral_constant<int, 6>{}); break;
            case 7: consume(std::integral_constant<int, 7>{}); break;
            case 8: consume(std::integral_constant<int, 8>{}); break;
            case 9: consume(std::integral_constant<int, 9>{}); break;
            case 10: consume(std::integral_constant<int, 10>{}); break;
            case 11: consume(std::integral_constant<int, 11>{}); break;
            case 12: consume(std::integral_constant<int, 12>{}); break;
            case 13: consume(std::integral_constant<int, 13>{}); break;
            case 14: consume(std::integral_constant<int, 14>{}); break;
            case 15: consume(std::integral_constant<int, 15>{}); break;
            case 16: consume(std::integral_constant<int, 16>{}); break;
            default: consume(std::integral_constant<int, 17>{}); break;
        }
    }
}

}  // namespace icv

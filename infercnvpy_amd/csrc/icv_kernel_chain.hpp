// Reference-order column sums (reference tl/_infercnv.py:385, :400: `np.mean(X, axis=0)` of the matrix as stored).
//
// numpy reduces a C-contiguous cells x genes matrix over axis 0 row by row: every column is ONE sequential chain
// `acc = fl(acc + x[r][g])` in the matrix dtype (float32 stays float32), rows ascending, then `acc / n`.  scipy's CSR
// mean is the same chain over `fl(x * fl(1/n))` (csc_matvecs of the transposed matrix walks the rows in order).  A
// float32 chain is not associative, so the only parallelism an exact reproduction has is ACROSS columns:
//
//   k_colchain<T>: one 1024-thread workgroup per tile of 512 bytes of a row (128 float32 / 64 float64 columns).
//     15 loader wavefronts copy the tile's row segments HBM -> LDS with LDS-DMA (`global_load_lds_dwordx4`: no staging
//     registers, no ds_write), kChDepth rounds of 30 rows ahead -- the LDS ring IS the set of bytes in flight (135 KB
//     per CU); ONE chain wavefront (lane = 2 float32 columns / 1 float64 column) adds the landed rows in order with
//     v_pk_add_f32 / v_add_f64, 8-byte conflict-free LDS reads.  One s_barrier per round hands a ring slot over.
//     The chain costs ~5 cycles per row (0.25 ms per 100 000 rows), far below the stream time of the tile, so the
//     kernel runs at the rate the loaders pull HBM.
//   CSR (k_colchain_csr): the same consumer; the loaders rebuild 64-row slices of the tile in LDS (zeros + the row's
//     stored entries of the tile's columns times 1/n); adding the zeros of the absent entries is exact (acc + 0 = acc).
//     Which entries of a row belong to a tile comes from k_csr_tile_bounds (one pass over the column indices).
//   CSC input (k_colpair_csc): scipy reduces a CSC matrix per column with np.add.reduceat = first entry + numpy's
//     pairwise sum of the rest; one thread per column restates that tree.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "icv_kernels.hpp"  // make_rsrc, u32x4

namespace icv {

constexpr int kChThreads = 1024;
constexpr int kChLoaders = 15;   // wavefronts 0..14 load, wavefront 15 adds
constexpr int kChMaxLines = 4;   // a tile is 1..4 cache lines (128 B) of a row: 64 chain lanes x 8 bytes at most
constexpr int kChLdsFull = 160 * 1024, kChLdsHalf = 80 * 1024;  // one / two workgroups per CU

typedef float chain_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void chain_add(chain_f2& acc, const chain_f2 v) {
    // two columns per lane in one issue slot; IEEE round-to-nearest per component, denormals kept
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(v));
}
__device__ __forceinline__ void chain_add(double& acc, const double v) {
    asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(v));
}

template <typename T> struct ChainLane;
template <> struct ChainLane<float> { typedef chain_f2 type; };
template <> struct ChainLane<double> { typedef double type; };

// LDS-DMA: 16 bytes per lane from `src` (per lane) to lds_base + 16 * lane (wave-uniform base, LDS byte address)
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_base)
        : "memory");
}

// Geometry of a tile of NL cache lines: LDS-DMA lanes per row, rows per 1 KB load, rows per round, bytes per ring slot
// (u = loads per loader wavefront and round: 2 with the whole LDS of a CU, 1 with half of it)
struct ChainGeom {
    int lpr, rpl, round_rows, row_bytes, slot_bytes;
    __host__ __device__ ChainGeom(int nl, int u)
        : lpr(8 * nl), rpl(64 / (8 * nl)), round_rows(u * kChLoaders * (64 / (8 * nl))), row_bytes(128 * nl),
          slot_bytes(u * kChLoaders * (64 / (8 * nl)) * 128 * nl) {}
};

// wait until at most n of this wavefront's LDS-DMA loads are outstanding (n wave-uniform, <= 15: the ring depth)
__device__ __forceinline__ void chain_wait_vmcnt(int n) {
    switch (n) {
#define ICV_CH_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        ICV_CH_W(1) ICV_CH_W(2) ICV_CH_W(3) ICV_CH_W(4) ICV_CH_W(5) ICV_CH_W(6) ICV_CH_W(7) ICV_CH_W(8)
        ICV_CH_W(9) ICV_CH_W(10) ICV_CH_W(11) ICV_CH_W(12) ICV_CH_W(13) ICV_CH_W(14) ICV_CH_W(15)
#undef ICV_CH_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// The chain wavefront's loop for a tile of NL lines, RR rows per round (= ring slot); lane = 8 bytes of the row (2 float32 /
// 1 float64 columns).  Hand-scheduled in chunks of ten rows (the asm operand limit is 30): the ten 8-byte LDS reads of
// the NEXT ten rows are issued, then the ten dependent adds of the current ten run in their shadow, one lgkmcnt wait per
// chunk; plain ds_read_b64 (the compiler's merged ds_read2st64_b64 costs four times the LDS cycles).  A lone wavefront
// issues a ds_read_b64 every ~10 cycles, which is what a row costs here: 0.6 ms per 100 000 rows, half the stream time.
// Measured alternatives (profiles/r04_colchain_experiments.txt): a read between every two adds, reads two chunks ahead in
// fixed registers with counted waits -- both slower on narrow tiles (the reads' issue, not their latency, is the cost).
#define ICV_CH_RD(i, RBS) "ds_read_b64 %[n" #i "], %[p] offset:" #i "*" RBS "\n\t"
#define ICV_CH_AD(OP, i) OP " %[a], %[a], %[c" #i "]\n\t"
#define ICV_CH_RD10(RBS)                                                                                         \
    ICV_CH_RD(0, RBS) ICV_CH_RD(1, RBS) ICV_CH_RD(2, RBS) ICV_CH_RD(3, RBS) ICV_CH_RD(4, RBS) ICV_CH_RD(5, RBS) \
    ICV_CH_RD(6, RBS) ICV_CH_RD(7, RBS) ICV_CH_RD(8, RBS) ICV_CH_RD(9, RBS)
#define ICV_CH_AD10(OP)                                                                                    \
    ICV_CH_AD(OP, 0) ICV_CH_AD(OP, 1) ICV_CH_AD(OP, 2) ICV_CH_AD(OP, 3) ICV_CH_AD(OP, 4) ICV_CH_AD(OP, 5) \
    ICV_CH_AD(OP, 6) ICV_CH_AD(OP, 7) ICV_CH_AD(OP, 8) ICV_CH_AD(OP, 9)
#define ICV_CH_OUT(NX)                                                                              \
    [n0] "=&v"(NX[0]), [n1] "=&v"(NX[1]), [n2] "=&v"(NX[2]), [n3] "=&v"(NX[3]), [n4] "=&v"(NX[4]), \
        [n5] "=&v"(NX[5]), [n6] "=&v"(NX[6]), [n7] "=&v"(NX[7]), [n8] "=&v"(NX[8]), [n9] "=&v"(NX[9])
#define ICV_CH_IN(CX)                                                                                      \
    [c0] "v"(CX[0]), [c1] "v"(CX[1]), [c2] "v"(CX[2]), [c3] "v"(CX[3]), [c4] "v"(CX[4]), [c5] "v"(CX[5]), \
        [c6] "v"(CX[6]), [c7] "v"(CX[7]), [c8] "v"(CX[8]), [c9] "v"(CX[9])
// ten rows at LDS address PX into NX[], waited for
#define ICV_CH_LOAD(RBS, NX, PX) asm volatile(ICV_CH_RD10(RBS) "s_waitcnt lgkmcnt(0)" : ICV_CH_OUT(NX) : [p] "v"(PX))
// the next ten rows into NX[] while the ten rows in CX[] are added
#define ICV_CH_STEP(RBS, OP, AX, NX, CX, PX)                                  \
    asm volatile(ICV_CH_RD10(RBS) ICV_CH_AD10(OP) "s_waitcnt lgkmcnt(0)"      \
                 : [a] "+v"(AX), ICV_CH_OUT(NX)                               \
                 : [p] "v"(PX), ICV_CH_IN(CX))
#define ICV_CH_ADDS(OP, AX, CX) asm volatile(ICV_CH_AD10(OP) : [a] "+v"(AX) : ICV_CH_IN(CX))

// FLAGS = false: a workgroup barrier hands every round over (the loaders of k_colchain run in lock step with it).
// FLAGS = true (k_colchain_csr): ready[slot] == k + 1 says round k is in its slot, *consumed = k + 1 gives it back --
// LDS words polled with workgroup-scope acquire loads, so that the producers work ahead of each other.
template <typename T, int NL, int RR, bool FLAGS = false>
__device__ __forceinline__ void chain_rounds(const unsigned char* smem, int n_slots, int64_t n_rounds, int64_t n_sel,
                                             int lane, typename ChainLane<T>::type& a, int* ready = nullptr,
                                             int* consumed = nullptr) {
    typedef typename ChainLane<T>::type lane_t;
    constexpr int RB = 128 * NL, SLOT = RR * RB;
    constexpr int NCH = RR / 10;
    static_assert(RR % 10 == 0, "round rows");
    int slot_i = 0;
    const unsigned lds0 = (unsigned)(uintptr_t)smem + (unsigned)lane * 8u;
    for (int64_t k = 0; k < n_rounds; ++k) {
        if constexpr (FLAGS) {
            while (__hip_atomic_load(ready + slot_i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)(k + 1))
                __builtin_amdgcn_s_sleep(1);
        } else {
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");  // the slot was written by other wavefronts (DMA / ds_write): read it now
        const unsigned p0 = lds0 + (unsigned)slot_i * (unsigned)SLOT;
        const lane_t* slot = reinterpret_cast<const lane_t*>(smem + (size_t)slot_i * SLOT) + lane;
        slot_i = slot_i + 1 == n_slots ? 0 : slot_i + 1;
        const int64_t left = n_sel - k * RR;
        if (left >= RR) {
            lane_t va[10], vb[10];
#define ICV_CH_BODY(RBS, OP)                                            \
    ICV_CH_LOAD(RBS, va, p0);                                           \
    _Pragma("unroll") for (int c = 0; c < NCH; c += 2) {                \
        if (c + 1 < NCH) {                                              \
            const unsigned p1 = p0 + (unsigned)((c + 1) * 10 * RB);     \
            ICV_CH_STEP(RBS, OP, a, vb, va, p1);                        \
            if (c + 2 < NCH) {                                          \
                const unsigned p2 = p0 + (unsigned)((c + 2) * 10 * RB); \
                ICV_CH_STEP(RBS, OP, a, va, vb, p2);                    \
            } else {                                                    \
                ICV_CH_ADDS(OP, a, vb);                                 \
            }                                                           \
        } else {                                                        \
            ICV_CH_ADDS(OP, a, va);                                     \
        }                                                               \
    }
            if constexpr (sizeof(T) == 4) {
                if constexpr (NL == 1) { ICV_CH_BODY("128", "v_pk_add_f32") }
                else if constexpr (NL == 2) { ICV_CH_BODY("256", "v_pk_add_f32") }
                else if constexpr (NL == 3) { ICV_CH_BODY("384", "v_pk_add_f32") }
                else { ICV_CH_BODY("512", "v_pk_add_f32") }
            } else {
                if constexpr (NL == 1) { ICV_CH_BODY("128", "v_add_f64") }
                else if constexpr (NL == 2) { ICV_CH_BODY("256", "v_add_f64") }
                else if constexpr (NL == 3) { ICV_CH_BODY("384", "v_add_f64") }
                else { ICV_CH_BODY("512", "v_add_f64") }
            }
#undef ICV_CH_BODY
        } else {
            for (int i = 0; i < (int)left; ++i) chain_add(a, slot[i * (RB / 8)]);
        }
        if constexpr (FLAGS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the round's reads have returned)
            if (lane == 0) __hip_atomic_store(consumed, (int)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// The chain wavefront of k_colchain_csr: round k sits in ring slot k % n_slots as nch[slot] chunks of ten LDS rows (the
// producer merges input rows that share no column of the tile into one LDS row, so a round of 60 input rows is 1..6
// chunks); ready[slot] == k + 1 says the round is there, *consumed = k + 1 gives the slot back.
template <typename T, int NL>
__device__ __forceinline__ void chain_rounds_var(const unsigned char* smem, int slot_bytes, int n_slots, int64_t n_rounds,
                                                 int lane, typename ChainLane<T>::type& a, int* ready, int* nch,
                                                 int* consumed) {
    typedef typename ChainLane<T>::type lane_t;
    constexpr int RB = 128 * NL;
    int slot_i = 0;
    const unsigned lds0 = (unsigned)(uintptr_t)smem + (unsigned)lane * 8u;
    for (int64_t k = 0; k < n_rounds; ++k) {
        while (__hip_atomic_load(ready + slot_i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)(k + 1))
            __builtin_amdgcn_s_sleep(1);
        const int n_chunks = __builtin_amdgcn_readfirstlane(nch[slot_i]);
        asm volatile("" ::: "memory");  // the slot was written by other wavefronts: read it now
        const unsigned p0 = lds0 + (unsigned)slot_i * (unsigned)slot_bytes;
        slot_i = slot_i + 1 == n_slots ? 0 : slot_i + 1;
        lane_t va[10], vb[10];
#define ICV_CH_VBODY(RBS, OP)                                              \
    ICV_CH_LOAD(RBS, va, p0);                                              \
    int c = 0;                                                             \
    for (; c + 2 < n_chunks; c += 2) {                                     \
        const unsigned p1 = p0 + (unsigned)((c + 1) * 10 * RB);            \
        ICV_CH_STEP(RBS, OP, a, vb, va, p1);                               \
        const unsigned p2 = p0 + (unsigned)((c + 2) * 10 * RB);            \
        ICV_CH_STEP(RBS, OP, a, va, vb, p2);                               \
    }                                                                      \
    if (c + 1 < n_chunks) { /* two chunks left: va loaded, c + 1 pending */ \
        const unsigned p1 = p0 + (unsigned)((c + 1) * 10 * RB);            \
        ICV_CH_STEP(RBS, OP, a, vb, va, p1);                               \
        ICV_CH_ADDS(OP, a, vb);                                            \
    } else {                                                               \
        ICV_CH_ADDS(OP, a, va);                                            \
    }
        if constexpr (sizeof(T) == 4) {
            if constexpr (NL == 1) { ICV_CH_VBODY("128", "v_pk_add_f32") }
            else if constexpr (NL == 2) { ICV_CH_VBODY("256", "v_pk_add_f32") }
            else if constexpr (NL == 3) { ICV_CH_VBODY("384", "v_pk_add_f32") }
            else { ICV_CH_VBODY("512", "v_pk_add_f32") }
        } else {
            if constexpr (NL == 1) { ICV_CH_VBODY("128", "v_add_f64") }
            else if constexpr (NL == 2) { ICV_CH_VBODY("256", "v_add_f64") }
            else if constexpr (NL == 3) { ICV_CH_VBODY("384", "v_add_f64") }
            else { ICV_CH_VBODY("512", "v_add_f64") }
        }
#undef ICV_CH_VBODY
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the round's reads have returned)
        if (lane == 0) __hip_atomic_store(consumed, (int)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// acc[c] (matrix dtype, in/out: a call continues the chain of the previous one) += rows sel[0..n_sel) of the dense
// row-major matrix x (sel == nullptr: rows 0..n_sel), in that order.  Workgroup b of gridDim.x owns the cache lines
// [b * n_lines / grid, (b + 1) * n_lines / grid) of every row (1..4 lines: the caller sizes the grid); `lds_bytes` of
// dynamic LDS (kChLdsFull / kChLdsHalf) are the ring.
// tail_row >= 0: the one row whose 16-byte segment loads could run past the end of the buffer (its last row, when the
// row stride is not a multiple of 16 bytes): it is added last by the chain wavefront with guarded loads.  Without a row
// list the caller excludes it from n_sel; with one the kernel looks whether the (ascending) list ends with it.
template <typename T, bool LIST>
__global__ void __launch_bounds__(kChThreads) k_colchain(const T* __restrict__ x, int64_t ld, int n_cols, int n_lines,
                                                         int lds_bytes, const int32_t* __restrict__ sel, int64_t n_sel,
                                                         int64_t tail_row, T* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (LIST && tail_row >= 0) {  // (uniform: kernel arguments and one scalar load)
        if ((int64_t)sel[n_sel - 1] == tail_row) n_sel -= 1;
        else tail_row = -1;
    }
    typedef typename ChainLane<T>::type lane_t;
    constexpr int EPL = 16 / (int)sizeof(T);  // elements per loader lane (one 16-byte load)
    constexpr int CPL = 8 / (int)sizeof(T);   // columns per chain lane
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int line0 = (int)((int64_t)blockIdx.x * n_lines / gridDim.x);
    const int nl = (int)((int64_t)(blockIdx.x + 1) * n_lines / gridDim.x) - line0;
    const int c0 = line0 * (128 / (int)sizeof(T));
    const int u = lds_bytes > kChLdsHalf ? 2 : 1;  // (uniform over the grid)
    const ChainGeom g(nl, u);
    const int n_slots = lds_bytes / g.slot_bytes;
    int depth = n_slots - 1;  // rounds in flight
    if (depth * u > 15) depth = 15 / u;
    const int64_t n_rounds = (n_sel + g.round_rows - 1) / g.round_rows;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;  // LDS byte address of the ring

    if (wave < kChLoaders) {
        // ---- loaders: lanes [q * lpr, (q + 1) * lpr) copy row q of the load's rpl rows; the rest idle ------------
        const int q = lane / g.lpr;
        int col = c0 + (lane - q * g.lpr) * EPL;
        if (col >= n_cols) col = 0;  // lanes past the last column: any valid address, their LDS bytes are never added
        // (a segment that straddles n_cols reads on into the row's padding or the next row: in bounds except on the
        // buffer's last row -- the caller's tail_row)
        const bool active = q < g.rpl;
        int issue_slot = 0;
        // LIST: the rows of one load are consecutive list entries, fetched with scalar loads (a vector load would have
        // the compiler wait for vmcnt(0) and drain the DMA queue) ONE ROUND AHEAD of their use, selected per lane
        int32_t rs_next[2] = {0, 0};
        const auto fetch_list = [&](int64_t round) {
            if (LIST && round < n_rounds) {
                for (int j = 0; j < u; ++j) {
                    const int64_t r0 = round * g.round_rows + (j * kChLoaders + wave) * g.rpl;
                    int32_t rs = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < g.rpl) {
                            const int64_t ri = r0 + i < n_sel ? r0 + i : n_sel - 1;
                            const int32_t v = sel[ri];
                            if (q == i) rs = v;
                        }
                    rs_next[j] = rs;
                }
            }
        };
        const auto issue = [&](int64_t round) {
            const int32_t rs0 = rs_next[0], rs1 = rs_next[1];
            for (int j = 0; j < u; ++j) {
                int64_t r = round * g.round_rows + (j * kChLoaders + wave) * g.rpl + q;
                if (r >= n_sel) r = n_sel - 1;
                if (LIST) r = j ? rs1 : rs0;
                const unsigned dst = lds0 + (unsigned)issue_slot * (unsigned)g.slot_bytes +
                                     (unsigned)((j * kChLoaders + wave) * g.rpl) * (unsigned)g.row_bytes;
                if (active) lds_dma16(x + r * ld + col, dst);
            }
            issue_slot = issue_slot + 1 == n_slots ? 0 : issue_slot + 1;
            fetch_list(round + 1);
        };
        fetch_list(0);
        for (int64_t k = 0; k < depth - 1 && k < n_rounds; ++k) issue(k);
        for (int64_t k = 0; k < n_rounds; ++k) {
            if (k + depth - 1 < n_rounds) {
                issue(k + depth - 1);
                chain_wait_vmcnt((depth - 1) * u);  // round k has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // ---- the chain: lane owns CPL adjacent columns ------------------------------------------------------------
        const int col = lane * 8 < g.row_bytes ? c0 + lane * CPL : n_cols;
        lane_t a;
        if constexpr (sizeof(T) == 4) {
            a.x = col < n_cols ? acc[col] : 0.f;
            a.y = col + 1 < n_cols ? acc[col + 1] : 0.f;
        } else {
            a = col < n_cols ? acc[col] : 0.0;
        }
        const int rl = lane * 8 < g.row_bytes ? lane : 0;  // idle lanes read lane 0's bytes (never stored)
        constexpr int L = kChLoaders;  // rows per round = loads per round (u x 15) x rows per load (64 / lanes per row)
        if (u == 2) {
            if (nl == 1) chain_rounds<T, 1, 2 * L * 8>(smem, n_slots, n_rounds, n_sel, rl, a);
            else if (nl == 2) chain_rounds<T, 2, 2 * L * 4>(smem, n_slots, n_rounds, n_sel, rl, a);
            else if (nl == 3) chain_rounds<T, 3, 2 * L * 2>(smem, n_slots, n_rounds, n_sel, rl, a);
            else chain_rounds<T, 4, 2 * L * 2>(smem, n_slots, n_rounds, n_sel, rl, a);
        } else {
            if (nl == 1) chain_rounds<T, 1, L * 8>(smem, n_slots, n_rounds, n_sel, rl, a);
            else if (nl == 2) chain_rounds<T, 2, L * 4>(smem, n_slots, n_rounds, n_sel, rl, a);
            else if (nl == 3) chain_rounds<T, 3, L * 2>(smem, n_slots, n_rounds, n_sel, rl, a);
            else chain_rounds<T, 4, L * 2>(smem, n_slots, n_rounds, n_sel, rl, a);
        }
        if (tail_row >= 0) {
            const T* xr = x + tail_row * ld;
            if constexpr (sizeof(T) == 4) {
                chain_f2 v;
                v.x = col < n_cols ? xr[col] : 0.f;
                v.y = col + 1 < n_cols ? xr[col + 1] : 0.f;
                chain_add(a, v);
            } else {
                chain_add(a, col < n_cols ? xr[col] : 0.0);
            }
        }
        if constexpr (sizeof(T) == 4) {
            if (col < n_cols) acc[col] = a.x;
            if (col + 1 < n_cols) acc[col + 1] = a.y;
        } else {
            if (col < n_cols) acc[col] = a;
        }
    }
}

// grid and LDS of k_colchain for n_cols columns of `esz` bytes on a device with n_cu compute units
struct ChainLaunch {
    int n_lines, grid, lds_bytes;
    ChainLaunch(int n_cols, int esz, int n_cu) {
        n_lines = (int)(((int64_t)n_cols * esz + 127) / 128);
        if (n_lines <= kChMaxLines * n_cu) {
            grid = n_lines < n_cu ? n_lines : n_cu;  // one workgroup per CU, the whole LDS each
            lds_bytes = kChLdsFull;
        } else {
            grid = (n_lines + kChMaxLines - 1) / kChMaxLines;
            lds_bytes = grid <= n_cu ? kChLdsFull : kChLdsHalf;
        }
        if (grid < 1) grid = 1;
    }
};


// ---------------------------------------------------------------------------------------------------------------------
// CSR input: the same chain over slices of the tile rebuilt in LDS
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kCcRows = 60;       // rows per round (one loader wavefront, lane = row; a multiple of the chain's ten)
constexpr int kCcBlock = 32;      // rows per workgroup of k_csr_tile_bounds
// Layout of the bounds table.  Row-major (a row's n_tiles + 1 words contiguous): a wavefront writes its row's words into
// ~1 KB (nine cache lines it completes itself), and the 32 tiles of an XCD read neighbouring words of ONE line per row
// (the tiles are dealt to the XCDs in contiguous runs).  The first layout, blocks of 32 rows x (tile, row): every store
// of a row went to a different line that 31 other rows had to complete -- 67 MB of open lines over the chip, 1.47 GB
// written for a 0.51 GB table.
#ifndef ICV_CC_TABLE_ROW_MAJOR
#define ICV_CC_TABLE_ROW_MAJOR 1
#endif
__host__ __device__ inline int64_t cc_bounds_index(int64_t i, int t, int n_tiles) {
#if ICV_CC_TABLE_ROW_MAJOR
    return i * (int64_t)(n_tiles + 1) + t;
#else
    return ((i / kCcBlock) * (int64_t)(n_tiles + 1) + t) * kCcBlock + (i % kCcBlock);
#endif
}

// bounds[cc_bounds_index(i, t)] = number of entries of selected row i in the tiles before t
// (t = 0 .. n_tiles; the row's entries of tile t are [bounds[t], bounds[t + 1]) from the row's start): one pass over the
// column indices, rows' columns ascending (canonical CSR).  line_tile[l] = tile of cache line l of a row.
// grid = ceil(n_sel / 32) workgroups of four wavefronts; a wavefront takes one row at a time.
template <bool LIST>
__global__ void __launch_bounds__(256) k_csr_tile_bounds(const int64_t* __restrict__ indptr,
                                                         const int32_t* __restrict__ indices,
                                                         const int32_t* __restrict__ sel, int64_t n_sel,
                                                         const uint16_t* __restrict__ line_tile, int esz_shift,
                                                         int n_tiles, uint32_t* __restrict__ bounds) {
    // (measured and dropped: the block's table staged in LDS and written out coalesced -- 64 KB of LDS leave two
    // workgroups per CU and the pass is bound by the latency of its dependent loads: 3.9 instead of 2.4 ms per 500 000
    // rows; four chunks of 64 entries in flight per wavefront instead)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < kCcBlock; r += 4) {
        const int64_t i = (int64_t)blockIdx.x * kCcBlock + r;
        if (i >= n_sel) break;  // (uniform)
        const int64_t row = LIST ? sel[i] : i;
        const int64_t e0 = indptr[row];
        const int64_t len = indptr[row + 1] - e0;
        int carry = -1;  // tile of the entry before this chunk
        for (int64_t base = 0; base <= len; base += 256) {
            int col[4], T[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t pos = base + u * 64 + lane;
                col[u] = pos < len ? indices[e0 + pos] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t pos = base + u * 64 + lane;
                // pos == len: the terminator closes every remaining tile at `len`
                T[u] = pos < len ? (int)line_tile[((unsigned)col[u] << esz_shift) >> 7] : n_tiles;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t pos = base + u * 64 + lane;
                int P = __shfl_up(T[u], 1);
                if (lane == 0) P = carry;
                carry = __shfl(T[u], 63);
                if (pos <= len)
                    for (int t = P + 1; t <= T[u]; ++t) bounds[cc_bounds_index(i, t, n_tiles)] = (uint32_t)pos;
            }
        }
    }
}

// acc[c] += fl(x * scale) over the stored entries of rows sel[0..n_sel) (nullptr: rows 0..n_sel), rows ascending --
// scipy's CSR mean(axis=0): sum over rows of x * (1/n) in the matrix dtype.  Tiles and ring as k_colchain (grid =
// ChainLaunch.grid = n_tiles of the bounds table); 15 loader wavefronts take the rounds in turn: the owner of round k
// (60 rows, lane = row) zeroes the slot and writes the rows' entries of the tile's columns (prefetched a turn earlier,
// their bounds two turns earlier) as soon as the chain wavefront has given the slot back.
template <typename T, bool LIST>
__global__ void __launch_bounds__(kChThreads) k_colchain_csr(const T* __restrict__ vals, const int64_t* __restrict__ indptr,
                                                             const int32_t* __restrict__ indices, int64_t n_rows_all,
                                                             const int32_t* __restrict__ sel, int64_t n_sel, int n_cols,
                                                             int n_lines, int lds_bytes,
                                                             const uint32_t* __restrict__ bounds, T scale,
                                                             T* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename ChainLane<T>::type lane_t;
    constexpr int CPL = 8 / (int)sizeof(T);
    constexpr int K = 64 / (int)sizeof(T);  // entries prefetched per row and tile (16 float32 / 8 float64)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // Workgroup b runs on XCD b % 8 (observed placement, used for speed only): XCD x takes a CONTIGUOUS eighth of the
    // tiles, so the 16-byte pieces its workgroups gather from a row's entry list share their cache lines in ONE L2 (with
    // tile = blockIdx the neighbours of a piece sat in eight different L2s: 45 GB fetched for 5.6 GB of entries, PMC)
    const int n_tiles = gridDim.x;
    const int xcd = blockIdx.x & 7, q_in_xcd = blockIdx.x >> 3;
    const int tile = xcd * (n_tiles >> 3) + (xcd < (n_tiles & 7) ? xcd : (n_tiles & 7)) + q_in_xcd;
    const int line0 = (int)((int64_t)tile * n_lines / n_tiles);
    const int nl = (int)((int64_t)(tile + 1) * n_lines / n_tiles) - line0;
    const int c0 = line0 * (128 / (int)sizeof(T));
    const int row_bytes = 128 * nl, slot_bytes = kCcRows * row_bytes;
    int n_slots = (lds_bytes - 512) / slot_bytes;
    if (n_slots > 32) n_slots = 32;  // (hand-over words: 32 slots)
    const int64_t n_rounds = (n_sel + kCcRows - 1) / kCcRows;
    // hand-over words behind the ring: ready[slot] = round in the slot + 1, consumed = rounds the chain is done with
    int* ready = reinterpret_cast<int*>(smem + (size_t)n_slots * slot_bytes);
    int* nch = ready + 32;       // chunks of ten LDS rows in the slot
    int* consumed = ready + 64;
    if (threadIdx.x <= 64) ready[threadIdx.x] = 0;
    __syncthreads();

    if (wave < kChLoaders) {
        const int64_t e_end = indptr[n_rows_all];
        // stage A: where the lane's row keeps its entries of this tile -- the three loaded words are kept RAW (the
        // arithmetic on them belongs to the next turn: written here, the compiler waits for them, and with them for
        // the eight entry loads issued just before, at the end of every turn: 2 us of exposed latency per round);
        // stage B: the first K entries
        int64_t a_rp = 0;
        uint32_t a_lo = 0, a_hi = 0;
        int64_t b_e = 0;
        int b_cnt = 0;
        int32_t b_idx[K];
        T b_val[K];
        const auto fetch_a = [&](int64_t k) {
            const int64_t i = k * kCcRows + lane;
            a_rp = 0;
            a_lo = a_hi = 0;
            if (k < n_rounds && lane < kCcRows && i < n_sel) {
                const int64_t row = LIST ? sel[i] : i;
                a_lo = bounds[cc_bounds_index(i, tile, n_tiles)];
                a_hi = bounds[cc_bounds_index(i, tile + 1, n_tiles)];
                a_rp = indptr[row];
            }
        };
        const auto fetch_b = [&]() {
            b_e = a_rp + (int64_t)a_lo;
            b_cnt = (int)(a_hi - a_lo);
            // range-checked 16-byte loads relative to the wavefront's first entry (rows ascend): lanes without entries
            // and reads past the end of the arrays return zeros
            int64_t e_first = b_cnt > 0 ? b_e : e_end;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const int64_t other = ((int64_t)__shfl_xor((int)(e_first >> 32), o) << 32) |
                                      (unsigned)__shfl_xor((int)e_first, o);
                e_first = other < e_first ? other : e_first;
            }
            e_first = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e_first >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)e_first);
            const int64_t left = e_end - e_first;
            const unsigned rec = (unsigned)(left < 0x3fffffff ? left : 0x3fffffff);
            const __amdgpu_buffer_rsrc_t i_rs = make_rsrc(indices + e_first, rec * 4u);
            const __amdgpu_buffer_rsrc_t v_rs = make_rsrc(vals + e_first, rec * (unsigned)sizeof(T));
            const int64_t rel = b_cnt > 0 ? b_e - e_first : 0x3fffffff;  // (no entries: out of range, zeros)
            const unsigned off = (unsigned)(rel < 0x3fffffff ? rel : 0x3fffffff);
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(i_rs, off * 4u, q * 16, 0);
                b_idx[4 * q] = (int)w.x, b_idx[4 * q + 1] = (int)w.y, b_idx[4 * q + 2] = (int)w.z, b_idx[4 * q + 3] = (int)w.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(v_rs, off * (unsigned)sizeof(T), q * 16, 0);
                if constexpr (sizeof(T) == 4) {
                    b_val[4 * q] = __uint_as_float(w.x), b_val[4 * q + 1] = __uint_as_float(w.y);
                    b_val[4 * q + 2] = __uint_as_float(w.z), b_val[4 * q + 3] = __uint_as_float(w.w);
                } else {
                    b_val[2 * q] = __hiloint2double((int)w.y, (int)w.x);
                    b_val[2 * q + 1] = __hiloint2double((int)w.w, (int)w.z);
                }
            }
        };
        // Input rows that share no column of the tile are MERGED into one LDS row: every column of the merged row gets at
        // most one addend, and adding the zeros of the other rows is exact, so the chain's sums do not change while its
        // work follows the stored entries (7 % density, 80-column tiles: ~1.6 input rows per LDS row).  Buddy scheme, all
        // lanes at once: rows 2i, 2i + 1 merge if their column sets (128-bit masks) are disjoint, two merged pairs
        // 4i .. 4i + 3 merge if the pairs' unions are disjoint; rows with more than K entries in the tile stay alone.  (A
        // sequential greedy walk over the lanes merges 2.3 rows on average but costs the producer 1.4 us of scalar lane
        // reads per round: measured 9.8 instead of 4.8 ms per 500 000 rows.)  Returns the chunks of ten LDS rows.
        const auto write_slot = [&](int64_t k) -> int {
            unsigned char* slot = smem + (size_t)(k % n_slots) * slot_bytes;
            const int64_t left = n_sel - k * kCcRows;
            const int rows_here = (int)(left < kCcRows ? left : kCcRows);
            unsigned long long m0 = 0, m1 = 0;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < b_cnt) {
                    const int c = b_idx[j] - c0;
                    if (c < 64) m0 |= 1ull << c;
                    else m1 |= 1ull << (c - 64);
                }
            if (b_cnt > K || lane >= rows_here) m0 = m1 = ~0ull;  // never merges (a lane past the rows: nothing to write)
            const auto xchg = [&](unsigned long long v, int o) {
                return ((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), o) << 32) | (unsigned)__shfl_xor((int)v, o);
            };
            const unsigned long long p0 = xchg(m0, 1), p1 = xchg(m1, 1);
            const bool pair = ((m0 & p0) | (m1 & p1)) == 0ull;               // (the same on both lanes of the pair)
            const unsigned long long u0 = m0 | p0, u1 = m1 | p1;
            const unsigned long long q0 = xchg(u0, 2), q1 = xchg(u1, 2);
            const bool pair_other = __shfl_xor((int)pair, 2) != 0;
            const bool quad = pair && pair_other && ((u0 & q0) | (u1 & q1)) == 0ull;
            const bool leader = lane < rows_here && (quad ? (lane & 3) == 0 : (pair ? (lane & 1) == 0 : true));
            const unsigned long long lead = __builtin_amdgcn_ballot_w64(leader);
            // LDS row of a lane = number of leaders at or before it, minus one
            const int my_g = __popcll(lead & ((2ull << lane) - 1ull)) - 1;
            const int g = __popcll(lead) - 1;
            const int n_lds_rows = g + 1;
            const int n_chunks = n_lds_rows > 0 ? (n_lds_rows + 9) / 10 : 1;
#if !(defined(ICV_DEV_EXPERIMENTS) && defined(ICV_CC_EXP_NOZERO))
            for (int o = lane * 16; o < n_chunks * 10 * row_bytes; o += 64 * 16)
                *reinterpret_cast<uint4*>(slot + o) = make_uint4(0, 0, 0, 0);
#endif
            T* row = reinterpret_cast<T*>(slot + (size_t)(my_g < 0 ? 0 : my_g) * row_bytes);
#if !(defined(ICV_DEV_EXPERIMENTS) && defined(ICV_CC_EXP_NOSCATTER))
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < b_cnt) row[b_idx[j] - c0] = b_val[j] * scale;
            for (int j = K; j < b_cnt; ++j) row[indices[b_e + j] - c0] = vals[b_e + j] * scale;  // long rows
#else
            if (b_cnt == 12345) row[b_idx[0] - c0] = b_val[3] * scale;
#endif
            return n_chunks;
        };
        // The producers are NOT in lock step with the chain (a barrier per round serialised their turns: 15 ms for
        // 500 000 rows, of which the chain needed 3.5): wavefront w fills the slots of rounds w, w + 15, ... as soon as
        // the chain has given the slot back, up to n_slots rounds ahead of it.
        int64_t mine = wave;  // this wavefront's next round
        fetch_a(mine);
        fetch_b();
        fetch_a(mine + kChLoaders);
        for (; mine < n_rounds; mine += kChLoaders) {
            const int need = (int)(mine - n_slots + 1);  // the chain is done with the round that had this slot
            while (__hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                __builtin_amdgcn_s_sleep(2);
            const int n_chunks = write_slot(mine);
            if (lane == 0) nch[(int)(mine % n_slots)] = n_chunks;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the slot's ds_writes have completed)
            if (lane == 0)
                __hip_atomic_store(ready + (int)(mine % n_slots), (int)(mine + 1), __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
#if !(defined(ICV_DEV_EXPERIMENTS) && defined(ICV_CC_EXP_NOFETCH))
            fetch_b();
            fetch_a(mine + 2 * kChLoaders);
#endif
        }
    } else {
        const int col = lane * 8 < row_bytes ? c0 + lane * CPL : n_cols;
        lane_t a;
        if constexpr (sizeof(T) == 4) {
            a.x = col < n_cols ? acc[col] : 0.f;
            a.y = col + 1 < n_cols ? acc[col + 1] : 0.f;
        } else {
            a = col < n_cols ? acc[col] : 0.0;
        }
        const int rl = lane * 8 < row_bytes ? lane : 0;
        if (nl == 1) chain_rounds_var<T, 1>(smem, slot_bytes, n_slots, n_rounds, rl, a, ready, nch, consumed);
        else if (nl == 2) chain_rounds_var<T, 2>(smem, slot_bytes, n_slots, n_rounds, rl, a, ready, nch, consumed);
        else if (nl == 3) chain_rounds_var<T, 3>(smem, slot_bytes, n_slots, n_rounds, rl, a, ready, nch, consumed);
        else chain_rounds_var<T, 4>(smem, slot_bytes, n_slots, n_rounds, rl, a, ready, nch, consumed);
        if constexpr (sizeof(T) == 4) {
            if (col < n_cols) acc[col] = a.x;
            if (col + 1 < n_cols) acc[col + 1] = a.y;
        } else {
            if (col < n_cols) acc[col] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// CSC input: scipy sums a CSC matrix over axis 0 with np.add.reduceat over each column's stored entries (after the
// multiplication by 1/n): first entry + numpy's pairwise sum of the others (8 accumulators up to 128 elements, halves
// rounded down to multiples of 8 above).  One thread per column walks its entries once per pass: the tree only needs
// the elements in order.  row_group != nullptr: only entries of rows with row_group[row] == group count (X[rows of the
// category, :] of a CSC matrix is again CSC).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct CscCursor {
    const T* vals;
    const int32_t* rows;
    const int32_t* row_group;
    int group;
    int64_t e, e1;
    T scale;
    __device__ bool skip() {  // -> positioned on the next counted entry (false: none left)
        if (row_group)
            while (e < e1 && row_group[rows[e]] != group) ++e;
        return e < e1;
    }
    __device__ T next() {
        skip();
        return vals[e++] * scale;
    }
};

template <typename T, typename Cursor>
__device__ T numpy_pairwise_stream(Cursor& c, int64_t n) {
    // explicit stack over numpy's recursion pairwise(a, n) = pairwise(a, n2) + pairwise(a + n2, n - n2)
    int64_t todo[48];
    T part[48];
    int state[48];
    int sp = 0;
    todo[0] = n;
    state[0] = 0;
    T ret = 0;
    while (sp >= 0) {
        const int64_t m = todo[sp];
        if (m <= 128) {
            T res;
            if (m < 8) {
                res = (T)0;
                for (int64_t i = 0; i < m; ++i) res = res + c.next();
            } else {
                T r[8];
                for (int q = 0; q < 8; ++q) r[q] = c.next();
                int64_t i = 8;
                for (; i < m - (m % 8); i += 8)
                    for (int q = 0; q < 8; ++q) r[q] = r[q] + c.next();
                res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < m; ++i) res = res + c.next();
            }
            ret = res;
            --sp;
            // return into the parents
            while (sp >= 0) {
                if (state[sp] == 1) {  // left half done: descend into the right half
                    part[sp] = ret;
                    state[sp] = 2;
                    int64_t n2 = todo[sp] / 2;
                    n2 -= n2 % 8;
                    ++sp;
                    todo[sp] = todo[sp - 1] - n2;
                    state[sp] = 0;
                    break;
                }
                ret = part[sp] + ret;  // state 2: both halves done
                --sp;
            }
        } else {
            state[sp] = 1;
            int64_t n2 = m / 2;
            n2 -= n2 % 8;
            ++sp;
            todo[sp] = n2;
            state[sp] = 0;
        }
    }
    return ret;
}

template <typename T>
__global__ void __launch_bounds__(64) k_colpair_csc(const T* __restrict__ vals, const int64_t* __restrict__ colptr,
                                                    const int32_t* __restrict__ rows, int n_cols,
                                                    const int32_t* __restrict__ row_group, int group, T scale,
                                                    T* __restrict__ out) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_cols) return;
    CscCursor<T> cur{vals, rows, row_group, group, colptr[c], colptr[c + 1], scale};
    int64_t n = cur.e1 - cur.e;
    if (row_group) {
        n = 0;
        for (int64_t e = cur.e; e < cur.e1; ++e) n += row_group[rows[e]] == group;
    }
    T res = (T)0;
    if (n > 0) {
        res = cur.next();
        if (n > 1) res = res + numpy_pairwise_stream<T>(cur, n - 1);
    }
    out[c] = res;
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense matrix stored column-major (np.asfortranarray, the transposed view of a genes x cells array): numpy's iterator
// puts the axis with the smaller stride innermost, so np.mean(X, axis=0) (reference :385) reduces every column with its
// contiguous inner loop -- the pairwise sum above -- over pieces of 8 192 elements (the iterator's buffer size; measured
// against numpy in the build container for float32 / float64, 1 .. 100 001 rows): sum = ((pw(x[0:8192]) + pw(x[8192:16384]))
// + ...).  One thread per column over the column's contiguous values (the caller uploads column blocks as they lie in
// host memory).  Not a hot path: it exists so that an F-ordered adata.X gives the reference's bits too.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct DenseCursor {
    const T* p;
    __device__ T next() { return *p++; }
};

constexpr int kNumpyBufferSize = 8192;

template <typename T>
__global__ void __launch_bounds__(64) k_colpair_dense(const T* __restrict__ xt, int64_t n, int n_cols, int64_t ld,
                                                      T* __restrict__ sums) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_cols) return;
    DenseCursor<T> cur{xt + (int64_t)c * ld};
    T res = (T)0;
    for (int64_t r = 0; r < n; r += kNumpyBufferSize) {
        const int64_t m = n - r < kNumpyBufferSize ? n - r : kNumpyBufferSize;
        const T part = numpy_pairwise_stream<T>(cur, m);
        res = r == 0 ? part : res + part;
    }
    sums[c] = res;
}

}  // namespace icv

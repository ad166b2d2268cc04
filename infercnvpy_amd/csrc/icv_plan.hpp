// Host-side plan: window table, padded LDS row layout, kernel variant selection.
// Pure host code (no HIP calls here) so that planning can be unit-tested without a GPU.
//
// Reference behaviour restated here (icbi-lab/infercnvpy, src/infercnvpy/tl/_infercnv.py):
//   :205-218  window < G_c  -> pyramid windows at gene offsets 0, step, 2*step, ...
//   :227-236  window >= G_c -> one flat window (plain mean) over the chromosome
//   :335-337  chr_pos[c] = cumulative window count
#pragma once
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

namespace icv {

constexpr int kThreads = 512;          // workgroup size of the smoothing kernel (8 wavefronts)
constexpr int kMaxBlocksPerThread = 8; // register-buffered partial sums (aliased LDS layout)
constexpr int kLdsLimit = 160 * 1024;  // gfx950: 160 KiB LDS per CU / per workgroup
constexpr int kScratchBytes = 1024;    // struct Scratch, rounded up
constexpr int kFastScratchBytes = 1536;  // struct ScratchF
constexpr int kFastUMax = 10;          // 16-byte loads per lane held in registers (fast path)
constexpr int kWsUMax = 12;            // same for the 448 worker threads of k_smooth_ws
constexpr int kWsMaxB = 5, kWsMaxW = 5, kWsMaxK = 30;
constexpr int kX16Threads = 1024;      // k_smooth_x16: 16 wavefronts, one workgroup per CU
constexpr int kX16UMax = 5;            // 16-byte row vectors per lane in flight

struct Layout {
    int elem_bytes = 4;
    int row_bytes = 0;   // padded row, rounded up to 16
    int s01_off = 0;     // blocked: partial sums {S0,S1} per block (aliases the row)
    int win_off = 0;     // float64 window values
    int scratch_off = 0;
    int total = 0;
    bool fits = false;
    // the float64 window array (8 W bytes) does not fit next to the row: k_smooth keeps it in a per-workgroup
    // HBM scratch line instead (L2 resident; e.g. 20 000 genes at step 1 = 17 822 windows)
    bool win_global = false;
};

struct Plan {
    int n_cols_all = 0, n_used = 0, n_chr = 0, window = 0, step = 0;
    int B = 1;      // genes per block; 1 = direct form
    int NB = 0;     // padded blocks (B > 1)
    int Gp = 0;     // padded row length (elements)
    int W = 0;      // windows
    std::vector<int32_t> chrom_off;  // n_chr + 1, sorted coordinates
    std::vector<int32_t> pad_off;    // n_chr + 1, padded coordinates
    std::vector<int32_t> chr_pos;    // n_chr, first window
    std::vector<int32_t> dst;        // n_cols_all: padded position or -1
    std::vector<int32_t> src;        // Gp: input column or -1 (pad)
    std::vector<int32_t> w_start;    // W: padded start position
    std::vector<int32_t> w_len;      // W: >0 pyramid length (genes); <0 flat over -len padded positions
    std::vector<double> w_denom;     // W: sum of weights / gene count
    std::vector<int32_t> w_start_sorted, w_len_sorted;  // sorted-gene coordinates, for the API
    std::vector<int32_t> pad_idx;    // padded positions without a gene
    // calculate_gene_values: genes covered by at least one kept window, in chromosome-sorted order
    std::vector<int32_t> cov_col, cov_j0, cov_cnt;  // input column, first covering window, #windows
    // the same coverage as RUNS of consecutive covered genes that share their windows (one value per run and cell: a
    // gene's value only depends on (j0, cnt)): run -> first window, #windows, #genes; input column -> run or -1
    std::vector<int32_t> gv_run_j0, gv_run_cnt, gv_run_mult, gv_col_run;
    std::vector<int32_t> w_pack;     // ws path: (start block & 0xffff) | (len << 16)
    // ws path, long windows (prefix-sum form): gene offset of a window inside its chromosome; per block, the gene
    // offset of its first gene inside its chromosome
    std::vector<int32_t> w_srel, blk_g0;
    std::vector<uint16_t> dst16;     // fast path: kFastUMax*kThreads*4 entries, Gp (trash slot) = masked
    double pyr_den = 1.0, pyr_rcp = 1.0;
    bool fast_ok = false;            // geometry admits k_smooth_fast (dense float32 input)
    bool ws_ok = false;              // ... and the wave-specialised k_smooth_ws
    // k_smooth_se (CSR float32 input: stored entries only): per window the LDS slots of its three prefix sums and the
    // wavefront totals to add, packed into two words (se_window_words)
    bool se_ok = false;
    std::vector<uint32_t> se_w0, se_w1;
    int fast_lds = 0, fast_scratch_off = 0, ws_win_off = 0, ws_hist_off = 0;
    // k_smooth_sp (one 1024-thread workgroup per CU): row | {S0,S1} | histogram | scratch, nothing aliased
    bool sp_ok = false;
    int sp_s01_off = 0, sp_hist_off = 0, sp_scratch_off = 0, sp_lds = 0;
    // k_smooth_x16 (one 1024-thread workgroup per CU, tables in registers): row | {S0,S1} | histogram | scratch
    bool x16_ok = false;
    int x16_s01_off = 0, x16_hist_off = 0, x16_scratch_off = 0, x16_lds = 0;
    int x16_half = 0;  // slots of the even-block {S0,S1} array (odd blocks follow)
    int x16_fine = 0;  // fine histogram bins: 4096, or 1024 where LDS is short (window 250: twice the blocks)
    std::vector<uint32_t> x16_wdesc;  // per thread: the pair of adjacent windows it owns (icv_kernels.hpp KParams)
    Layout lay32, lay64;
};

// k_smooth_se keeps the per-block sums {S0, T1} of a cell -- and then their prefix sums over the blocks of each
// WAVEFRONT (thread t of 512 owns blocks 8 t .. 8 t + 7, wavefront w blocks 512 w .. 512 w + 511) -- at slot
// (b & 7) * 513 + (b >> 3) of its LDS planes (16 bytes per slot; slot 512 of plane 0 is never written: zero).
// A window needs the prefix at the block before it (b-), at the last block of its first half (m1) and at its last
// block (m2); differences of prefixes of one wavefront are exact differences, and where m1 / m2 lie in the
// wavefront after that of b- the total of b-'s wavefront is added (a window is shorter than 512 blocks).
//   w0 = slot(b-) | slot(m1) << 13 | tm << 26 | flat << 31      tm / te: index into tot[16] of the total to add to the
//   w1 = slot(m2) | sg << 13 | te << 27                         prefix at m1 / m2: wavefront of b-, or 8 + it (zero)
// sg: gene offset of the window inside its chromosome (pyramid), or the gene count of the chromosome (flat window).
constexpr int kSePlane = 513;
constexpr int se_slot(int b) { return (b & 7) * kSePlane + (b >> 3); }
constexpr int kSeZeroSlot = 512;  // plane 0, slot 512
inline void se_window_words(int bs, int nbw, int hbw, bool flat, int flat_blocks, int sg, uint32_t& w0, uint32_t& w1) {
    const int wb = bs > 0 ? (bs - 1) >> 9 : 0;
    const int m1 = flat ? bs : bs + hbw - 1;
    const int m2 = bs + (flat ? flat_blocks : nbw) - 1;
    const uint32_t sb = bs > 0 ? (uint32_t)se_slot(bs - 1) : (uint32_t)kSeZeroSlot;
    const uint32_t tm = (uint32_t)(((m1 >> 9) != wb) ? wb : 8 + wb), te = (uint32_t)(((m2 >> 9) != wb) ? wb : 8 + wb);
    w0 = sb | ((uint32_t)se_slot(m1) << 13) | (tm << 26) | ((flat ? 1u : 0u) << 31);
    w1 = (uint32_t)se_slot(m2) | ((uint32_t)sg << 13) | (te << 27);
}

inline int gcd_int(int a, int b) { return b == 0 ? a : gcd_int(b, a % b); }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline Layout make_layout(const Plan& p, int elem_bytes) {
    Layout l;
    l.elem_bytes = elem_bytes;
    l.row_bytes = round_up(p.Gp * elem_bytes, 16);
    int data;
    if (p.B > 1) {
        l.s01_off = 0;
        l.win_off = 16 * p.NB;
        int alias = l.win_off + round_up(8 * p.W, 16);
        data = alias > l.row_bytes ? alias : l.row_bytes;
    } else {
        l.win_off = l.row_bytes;
        data = l.row_bytes + round_up(8 * p.W, 16);
    }
    l.scratch_off = round_up(data, 16);
    l.total = l.scratch_off + kScratchBytes;
    l.fits = l.total <= kLdsLimit;
    if (!l.fits) {  // retry without the window array in LDS
        int data2 = l.row_bytes;
        if (p.B > 1 && 16 * p.NB > data2) data2 = 16 * p.NB;
        const int total2 = round_up(data2, 16) + kScratchBytes;
        if (total2 <= kLdsLimit) {
            l.win_global = true;
            l.win_off = 0;  // unused
            l.scratch_off = round_up(data2, 16);
            l.total = total2;
            l.fits = true;
        }
    }
    return l;
}

// Returns "" on success, an error message otherwise.
// force_B > 0: use this block size (the chromosome groups of a split gene set evaluate their windows in the same
// canonical order as the full plan, whose window table the exact tie-break of the threshold kernel uses)
inline std::string build_plan(Plan& p, int n_cols_all, const int32_t* col_pos, int n_chr,
                              const int32_t* chrom_off, int window, int step, int force_B = 0) {
    if (n_cols_all <= 0) return "n_cols_all must be positive";
    if (n_chr <= 0) return "no chromosome with genes to smooth (need names starting with 'chr')";
    if (window < 1 || step < 1) return "window and step must be >= 1";
    p.n_cols_all = n_cols_all;
    p.n_chr = n_chr;
    p.window = window;
    p.step = step;
    p.chrom_off.assign(chrom_off, chrom_off + n_chr + 1);
    if (p.chrom_off[0] != 0) return "chrom_offsets[0] must be 0";
    for (int c = 0; c < n_chr; ++c)
        if (p.chrom_off[c + 1] <= p.chrom_off[c]) return "every chromosome needs at least one gene";
    p.n_used = p.chrom_off[n_chr];

    // col_pos must be a bijection onto [0, n_used)
    std::vector<int32_t> order(p.n_used, -1);
    for (int g = 0; g < n_cols_all; ++g) {
        int q = col_pos[g];
        if (q < 0) continue;
        if (q >= p.n_used || order[q] != -1) return "col_pos is not a permutation of the used genes";
        order[q] = g;
    }
    for (int q = 0; q < p.n_used; ++q)
        if (order[q] < 0) return "col_pos does not cover every used gene";

    // block size: pyramid weights are linear on [0, n/2) and [n/2, n); a block must not straddle
    // the kink, must tile the window and the step.
    int B = 1;
    if (window % 2 == 0) B = gcd_int(step, window / 2);
    auto count_blocks = [&](int b) {
        long nb = 0;
        for (int c = 0; c < n_chr; ++c) nb += (p.chrom_off[c + 1] - p.chrom_off[c] + b - 1) / b;
        return nb;
    };
    if (B > 1 && count_blocks(B) > (long)kThreads * kMaxBlocksPerThread) B = 1;
    if (force_B > 0) B = force_B;
    p.B = B;

    // padded layout + window table
    p.pad_off.assign(n_chr + 1, 0);
    p.chr_pos.assign(n_chr, 0);
    p.w_start.clear(); p.w_len.clear(); p.w_denom.clear();
    p.w_start_sorted.clear(); p.w_len_sorted.clear();
    int pad = 0, w = 0;
    for (int c = 0; c < n_chr; ++c) {
        int gc = p.chrom_off[c + 1] - p.chrom_off[c];
        int padded = round_up(gc, B);
        p.pad_off[c] = pad;
        p.chr_pos[c] = w;
        if (window < gc) {
            int nwin = (gc - window + 1 + step - 1) / step;
            double denom = (window % 2 == 0) ? (double)(window / 2) * (double)(window / 2 + 1)
                                             : (double)((window + 1) / 2) * (double)((window + 1) / 2);
            for (int j = 0; j < nwin; ++j) {
                p.w_start.push_back(pad + j * step);
                p.w_len.push_back(window);
                p.w_denom.push_back(denom);
                p.w_start_sorted.push_back(p.chrom_off[c] + j * step);
                p.w_len_sorted.push_back(window);
            }
            w += nwin;
        } else {
            p.w_start.push_back(pad);
            p.w_len.push_back(-padded);
            p.w_denom.push_back((double)gc);
            p.w_start_sorted.push_back(p.chrom_off[c]);
            p.w_len_sorted.push_back(gc);
            w += 1;
        }
        pad += padded;
    }
    p.pad_off[n_chr] = pad;
    p.Gp = pad;
    p.NB = (B > 1) ? pad / B : 0;
    p.W = w;

    p.src.assign(p.Gp, -1);
    p.dst.assign(n_cols_all, -1);
    for (int c = 0; c < n_chr; ++c) {
        int gc = p.chrom_off[c + 1] - p.chrom_off[c];
        for (int i = 0; i < gc; ++i) {
            int g = order[p.chrom_off[c] + i];
            p.src[p.pad_off[c] + i] = g;
            p.dst[g] = p.pad_off[c] + i;
        }
    }
    p.lay32 = make_layout(p, 4);
    p.lay64 = make_layout(p, 8);

    // per-gene window coverage (reference _calculate_gene_averages, :247-291): kept window j of a
    // chromosome covers sorted genes [j*step, j*step + window); small chromosomes: the single window
    p.cov_col.clear(); p.cov_j0.clear(); p.cov_cnt.clear();
    for (int c = 0; c < n_chr; ++c) {
        const int gc = p.chrom_off[c + 1] - p.chrom_off[c];
        const int w0 = p.chr_pos[c];
        const int wc = (c + 1 < n_chr ? p.chr_pos[c + 1] : p.W) - w0;
        for (int i = 0; i < gc; ++i) {
            int j0 = 0, j1 = 0;  // windows [j0, j1]
            if (window < gc) {
                j0 = (i - window + 1 + step - 1) / step;
                if (i - window + 1 < 0) j0 = 0;
                j1 = i / step;
                if (j1 > wc - 1) j1 = wc - 1;
                if (j1 < j0) continue;  // not covered -> NaN in the output
            }
            p.cov_col.push_back(order[p.chrom_off[c] + i]);
            p.cov_j0.push_back(w0 + j0);
            p.cov_cnt.push_back(j1 - j0 + 1);
        }
    }

    p.gv_run_j0.clear(); p.gv_run_cnt.clear(); p.gv_run_mult.clear();
    p.gv_col_run.assign((size_t)n_cols_all, -1);
    for (size_t q = 0; q < p.cov_col.size(); ++q) {
        if (p.gv_run_j0.empty() || p.gv_run_j0.back() != p.cov_j0[q] || p.gv_run_cnt.back() != p.cov_cnt[q]) {
            p.gv_run_j0.push_back(p.cov_j0[q]);
            p.gv_run_cnt.push_back(p.cov_cnt[q]);
            p.gv_run_mult.push_back(0);
        }
        ++p.gv_run_mult.back();
        p.gv_col_run[p.cov_col[q]] = (int32_t)p.gv_run_j0.size() - 1;
    }

    p.pad_idx.clear();
    for (int i = 0; i < p.Gp; ++i)
        if (p.src[i] < 0) p.pad_idx.push_back(i);
    p.pyr_den = (window % 2 == 0) ? (double)(window / 2) * (double)(window / 2 + 1)
                                  : (double)((window + 1) / 2) * (double)((window + 1) / 2);
    p.pyr_rcp = 1.0 / p.pyr_den;

    // packed window table {start block (16 bits) | length << 16} of the ws / se kernels; k_smooth_se: per window the
    // gene offset inside its chromosome, per block that of its first gene.  Needs every length to fit 16 signed bits.
    p.se_ok = false;
    bool pack_ok = B > 1 && window <= 32767 && p.NB <= 65535;
    for (int c = 0; c < n_chr && pack_ok; ++c) pack_ok = p.pad_off[c + 1] - p.pad_off[c] <= 32767;
    p.w_pack.clear(); p.w_srel.clear(); p.blk_g0.clear();
    if (pack_ok) {
        p.w_pack.resize(p.W);
        for (int j = 0; j < p.W; ++j)
            p.w_pack[j] = (int32_t)((uint32_t)((p.w_start[j] / B) & 0xffff) | ((uint32_t)p.w_len[j] << 16));
        p.w_srel.assign(p.W, 0);
        p.blk_g0.assign((size_t)p.NB + 8, 0);
        for (int c = 0; c < n_chr; ++c) {
            const int w0 = p.chr_pos[c], wc = (c + 1 < n_chr ? p.chr_pos[c + 1] : p.W) - w0;
            for (int q = 0; q < wc; ++q) p.w_srel[w0 + q] = p.w_start[w0 + q] - p.pad_off[c];
            for (int b = p.pad_off[c] / B; b < p.pad_off[c + 1] / B; ++b) p.blk_g0[b] = b * B - p.pad_off[c];
        }
        // k_smooth_se: block bins in 8 planes of 512 slots, four windows per thread of a 512-thread workgroup; the
        // cells it hands back go to the generic kernel (one row in LDS); windows shorter than a wavefront's 512
        // blocks, gene offsets that fit the 14-bit field
        p.se_ok = window % 2 == 0 && p.NB <= 8 * kThreads && p.W <= 4 * kThreads && p.lay32.fits && window / B <= 512;
        for (int c = 0; c < n_chr && p.se_ok; ++c) p.se_ok = p.pad_off[c + 1] - p.pad_off[c] < 16384;
        p.se_w0.clear(); p.se_w1.clear();
        if (p.se_ok) {
            p.se_w0.resize(p.W);
            p.se_w1.resize(p.W);
            const int nbw = window / B, hbw = nbw / 2;
            for (int j = 0; j < p.W; ++j) {
                const bool flat = p.w_len[j] < 0;
                se_window_words(p.w_start[j] / B, nbw, hbw, flat, flat ? -p.w_len[j] / B : 0,
                                flat ? (int)p.w_denom[j] : p.w_srel[j], p.se_w0[j], p.se_w1[j]);
            }
        }
    }

    // fast path: float32 dense, blocked form, row + tables fit the register/LDS budget
    p.fast_ok = false;
    if (pack_ok && n_cols_all % 4 == 0 && n_cols_all <= kFastUMax * kThreads * 4 && p.Gp < 65535 &&
        window <= 32767 && p.Gp <= 32767 * 1) {
        int data = round_up((p.Gp + 1) * 4, 16);  // + trash slot for masked columns
        if (16 * p.NB > data) data = 16 * p.NB;
        p.fast_scratch_off = round_up(data, 16);
        p.fast_lds = p.fast_scratch_off + kFastScratchBytes;
        p.fast_ok = p.fast_lds <= kLdsLimit;
        p.dst16.assign((size_t)kWsUMax * (kThreads - 64) * 4, (uint16_t)p.Gp);  // >= kFastUMax*kThreads*4
        p.ws_win_off = 16 * p.NB;
        p.ws_hist_off = p.ws_win_off;  // the histogram follows {S0,S1} inside the (dead) row
        p.ws_ok = p.fast_ok && p.ws_hist_off + 4096 * 2 <= p.fast_scratch_off && p.W < 65536 && p.NB <= kThreads * 8 &&
                  p.W <= kThreads * 4;
        for (int g = 0; g < n_cols_all; ++g)
            if (p.dst[g] >= 0) p.dst16[g] = (uint16_t)p.dst[g];
        p.sp_s01_off = round_up((p.Gp + 1) * 4, 16);
        p.sp_hist_off = p.sp_s01_off + 2 * 16 * p.NB;  // {S0,S1} double buffered
        p.sp_scratch_off = p.sp_hist_off + 4096 * 2;
        p.sp_lds = p.sp_scratch_off + kFastScratchBytes;
        p.sp_ok = p.ws_ok && p.sp_lds <= kLdsLimit && p.NB <= kThreads * 4 && p.W <= kThreads * 4;
        p.x16_s01_off = round_up((p.Gp + 1) * 4, 16);
        // {S0,S1} of block b: array b % inter, slot b / inter (inter = 2 * step / B: adjacent windows of a thread
        // are step / B blocks apart, threads 2 * step / B); the reads of a window pair may pass the last block
        const int sbk = (B > 0 && step % B == 0) ? step / B : 1, inter = 2 * sbk;
        p.x16_half = round_up((p.NB + window / B + 2 * sbk + inter) / inter + 1, 8);
        p.x16_hist_off = p.x16_s01_off + inter * 16 * p.x16_half;
        p.x16_ok = false;
        for (int fine : {4096, 1024}) {
            const int hist_bytes = 2 * (fine * 4 + 16 * (fine / 64) * 4);  // fine + 16 coarse replicas, both parities
            p.x16_fine = fine;
            p.x16_scratch_off = p.x16_hist_off + hist_bytes;
            p.x16_lds = p.x16_scratch_off + kFastScratchBytes + round_up(4 * p.W, 16);  // + x_res staging row
            if (p.x16_lds <= kLdsLimit) {
                p.x16_ok = true;
                break;
            }
        }
        p.x16_ok = p.x16_ok && p.fast_ok && n_cols_all <= kX16UMax * kX16Threads * 4 && p.NB <= 4080 && p.W <= 4095;
        // windows dealt to threads in pairs of adjacent windows of one chromosome
        p.x16_wdesc.assign(2 * kX16Threads, 0u);  // [threads] pair descriptors, then [threads] flat-window info
        {
            int idx = 0;
            for (int c = 0; c < n_chr && p.x16_ok; ++c) {
                const int w0 = p.chr_pos[c], wc = (c + 1 < n_chr ? p.chr_pos[c + 1] : p.W) - w0;
                for (int q = 0; q < wc; q += 2) {
                    if (idx >= kX16Threads) {
                        p.x16_ok = false;
                        break;
                    }
                    const int j0 = w0 + q;
                    const bool v1 = q + 1 < wc;
                    const uint32_t full0 = p.w_len[j0] == window ? 1u : 0u;
                    const uint32_t full1 = (v1 && p.w_len[j0 + 1] == window) ? 1u : 0u;
                    if (!full0) {  // flat window (the only one of its chromosome): blocks | gene count << 16
                        const int nbk = -p.w_len[j0] / B, gcn = (int)p.w_denom[j0];
                        if (v1 || nbk > 0xffff || gcn > 0x7fff) p.x16_ok = false;
                        p.x16_wdesc[kX16Threads + idx] = (uint32_t)nbk | ((uint32_t)gcn << 16);
                    }
                    p.x16_wdesc[idx++] = (uint32_t)(p.w_start[j0] / B) | ((uint32_t)j0 << 12) | (1u << 24) |
                                         ((v1 ? 1u : 0u) << 25) | (full0 << 26) | (full1 << 27);
                }
            }
        }
    }
    return "";
}

}  // namespace icv

// C ABI (include/infercnv_hip.h) over the gfx950 kernels.  No torch types, no exceptions across
// the boundary; all data pointers are device pointers owned by the caller.
#include "../../include/infercnv_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <memory>
#include <vector>

#include "icv_kernels.hpp"
#include "icv_kernel_ws.hpp"
#include "icv_kernel_x16.hpp"
#include "icv_kernel_se.hpp"
#include "icv_kernel_chain.hpp"
#include "icv_kernel_chainq.hpp"
#include "icv_kernel_pack.hpp"
#include "icv_kernel_util.hpp"
#include "icv_kernel_gene.hpp"
#include "icv_kernel_blocks.hpp"
#include "icv_corr.hpp"
#include "icv_ward.hpp"
#include "icv_ward_strip.hpp"
#include "icv_plan.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ICV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorName(e_) + " (" + \
                                         hipGetErrorString(e_) + ")");                         \
    } while (0)

// Developer knobs (environment variables that force a kernel route / geometry for A/B timing and for the tests that
// compare the kernels with each other).  They are read ONCE, at the first dispatch, not on every call (a
// getenv per dispatch, and a production call's route must not follow an environment edited under it); a test that
// changes them calls icv_developer_knobs_reload().
struct Knobs {
    bool force_generic, no_x16, no_sd, phase_profile, ward_in_place, no_mask_ring, no_fill_ring, no_chain_queues, se_maxw4,
         no_gene_fused;
    int wgs_per_cu;        // 0 = not set
    int chain_far;         // ICV_CHAIN_FAR: entries a buffer offset may span in k_colchain_csrq (tests: a small value)
    int chain_pieces;      // ICV_CHAIN_PIECES: lanes per row of k_colchain_csrq (2 / 4 / 8 / 16; 0 = from the density)
    double ward_compact_x; // 0 = not set
    void load() {
        force_generic = std::getenv("ICV_FORCE_GENERIC") != nullptr;
        no_x16 = std::getenv("ICV_NO_X16") != nullptr;
        no_sd = std::getenv("ICV_NO_SD") != nullptr;
        phase_profile = std::getenv("ICV_PHASE_PROFILE") != nullptr;
        ward_in_place = std::getenv("ICV_WARD_IN_PLACE") != nullptr;
        no_mask_ring = std::getenv("ICV_NO_MASK_RING") != nullptr;
        no_fill_ring = std::getenv("ICV_NO_FILL_RING") != nullptr;
        no_chain_queues = std::getenv("ICV_NO_CHAIN_QUEUES") != nullptr;
        se_maxw4 = std::getenv("ICV_SE_MAXW4") != nullptr;
        no_gene_fused = std::getenv("ICV_NO_GENE_FUSED") != nullptr;  // gene values through the three round-1 kernels
        const char* e = std::getenv("ICV_WGS_PER_CU");
        wgs_per_cu = e ? std::atoi(e) : 0;
        e = std::getenv("ICV_CHAIN_FAR");
        chain_far = e ? std::atoi(e) : 0;
        e = std::getenv("ICV_CHAIN_PIECES");
        chain_pieces = e ? std::atoi(e) : 0;
        e = std::getenv("ICV_WARD_COMPACT_X");
        ward_compact_x = e ? std::atof(e) : 0.0;
    }
};
std::mutex g_knobs_mu;
bool g_knobs_loaded = false;
Knobs g_knobs;
const Knobs& knobs() {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    if (!g_knobs_loaded) {
        g_knobs.load();
        g_knobs_loaded = true;
    }
    return g_knobs;
}

}  // namespace

struct icv_plan_s {
    icv::Plan p;
    std::mutex mu;
    std::atomic<bool> busy{false};  // a compute entry point is in its launch sequence (workspace in use)
    int device = -1;
    int n_cu = 0;
    int32_t *d_dst = nullptr, *d_src = nullptr, *d_wstart = nullptr, *d_wlen = nullptr;
    double* d_wdenom = nullptr;
    int32_t* d_pad = nullptr;
    int32_t* d_wpack = nullptr;
    unsigned* d_tie_n = nullptr;  // k_thr_mask_ring's tie counter + k_thr_mask_ties' done counter (zero between calls)
    int32_t *d_cov_col = nullptr, *d_cov_j0 = nullptr, *d_cov_cnt = nullptr;
    uint32_t* d_gv_pk = nullptr;  // k_gene_fused: per run, first window | #windows << 16
    int32_t* d_gv_mult = nullptr;   //               genes per run
    uint16_t* d_gv_col16 = nullptr;  //              input column -> run, or R (the kernel's NaN slot) where the gene has no value; padded to a multiple of 8 columns
    bool gv_fused_ok = false;
    int gv_mult_bytes = 2;  // bytes per run length in the kernel's LDS (1 where every run has at most 255 genes)
    int64_t* d_row_list = nullptr;  // cells handed back by k_smooth_ws to the generic kernel
    int* d_row_count = nullptr;
    int64_t row_list_cap = 0;
    // deferred profiling (icv_profile_begin / icv_profile_collect): event quadruples of the runs since begin
    bool prof_deferred = false;
    std::vector<hipEvent_t> prof_events;
    double* d_win_scratch = nullptr;  // Layout::win_global: per-workgroup window lines of k_smooth
    int64_t win_scratch_cap = 0;
    double* d_cell_part = nullptr;  // per-wavefront partial moments of the ws / x16 kernels
    int64_t cell_part_cap = 0;
    double* d_chunk_part = nullptr;  // k_smooth_x16<CHUNK>: per (chunk, workgroup, wavefront) moments
    int64_t chunk_part_cap = 0;      // in double2 elements
    double* d_hb_stats = nullptr;    // per-row moments when the caller passes no cell_stats (rows x 2)
    int64_t hb_stats_cap = 0;        // in rows
    uint16_t* d_dst16 = nullptr;
    uint32_t* d_x16_wdesc = nullptr;
    int32_t* d_blk_g0 = nullptr;  // per block: first-gene offset inside its chromosome (k_se_wtab)
    uint32_t *d_se_w0 = nullptr, *d_se_w1 = nullptr;   // k_smooth_se (plan: se_window_words)
    void* d_zrow = nullptr;  // CSR workspace: padded row, sized for float64
    size_t zrow_elems = 0;
    // Gene sets whose padded row does not fit LDS (float32: > ~40 000 genes, float64: > ~20 000): the
    // chromosomes are dealt into groups that do fit, each group is a plan of its own (windows only), and the
    // median / centring run on the float64 windows collected in HBM (smooth_split).
    std::vector<icv_plan_s*> parts;
    std::vector<int> part_woff;  // first window of every group in the full window list
    int parts_elem = 0;          // element size the groups were sized for (0: not built)
    int last_kernel = ICV_KERNEL_NONE;  // smoothing kernel of the last compute call (icv_plan_last_kernel)
    // the per-call workspace above is shared by all calls on this plan: a call on another stream than the previous
    // one first waits (on the device) for the previous call's work
    hipEvent_t done_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
};

namespace {

// grid of k_smooth_se (workgroups per CU from its LDS map; ICV_WGS_PER_CU: developer knob for occupancy experiments) --
// ONE definition for the launch and for the count of per-workgroup partial-moment slots the thresholds read (ADVICE r3:
// the two used to be computed separately and agreed only while kSeLds gave exactly two workgroups per CU)
int64_t se_grid(const icv_plan_t pl, int64_t n_rows) {
    int per_cu = icv::kLdsLimit / icv::kSeLds;
    if (const int v = knobs().wgs_per_cu)
        if (v >= 1 && v < per_cu) per_cu = v;
    int64_t grid = (int64_t)pl->n_cu * per_cu;
    return grid > n_rows ? n_rows : grid;
}

// one compute call at a time per plan (the plan owns the per-call workspace); released on scope exit
struct PlanBusy {
    icv_plan_t pl;
    bool ok;
    hipStream_t st = nullptr;
    bool entered = false;
    explicit PlanBusy(icv_plan_t p) : pl(p), ok(false) {
        bool expected = false;
        ok = p && p->busy.compare_exchange_strong(expected, true);
    }
    // Work of the previous call on this plan may still be in flight on ANOTHER stream and uses the same workspace
    // (row list, partial moments, zero row): this call's stream waits for it.  Same stream: already ordered.
    hipError_t enter(hipStream_t s) {
        st = s;
        entered = true;
        if (!pl->done_ev) {
            const hipError_t e = hipEventCreateWithFlags(&pl->done_ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        if (pl->has_last && pl->last_stream != s) return hipStreamWaitEvent(s, pl->done_ev, 0);
        return hipSuccess;
    }
    ~PlanBusy() {
        if (ok && entered && pl->done_ev) {
            if (hipEventRecord(pl->done_ev, st) == hipSuccess) {
                pl->last_stream = st;
                pl->has_last = true;
            }
        }
        if (ok) pl->busy.store(false);
    }
};
#define PLAN_BUSY_GUARD(pl)                                                                          \
    PlanBusy busy_guard_(pl);                                                                        \
    if (!busy_guard_.ok)                                                                             \
        return fail(ICV_ERR_INVALID, "plan busy: one compute call at a time per plan (see icv_plan_create)")
// after ensure_device: order this call behind the plan's previous call if that ran on another stream
#define PLAN_ENTER(stream_) HIP_TRY(busy_guard_.enter(static_cast<hipStream_t>(stream_)))

// stream-ordered temporary: freed on every exit path (error paths leaked their buffers)
struct AsyncBuf {
    void* p = nullptr;
    hipStream_t st = nullptr;
    ~AsyncBuf() {
        if (p) (void)hipFreeAsync(p, st);
    }
    hipError_t alloc(size_t bytes, hipStream_t s) {
        st = s;
        keep_pool();
        return hipMallocAsync(&p, bytes ? bytes : 16, s);
    }
    // The device's default memory pool gives everything back to the driver whenever the stream is synchronised (release
    // threshold 0), and the next temporary is a real allocation again: milliseconds of an idle GPU in front of a
    // 0.2 ms kernel for every caller that synchronises between calls.  Keep up to 6 GB of temporaries pooled (once
    // per device and process; the largest single temporary is the n x n correlation matrix of tl.ithcna: 2.5 GB at
    // 25 000 cells per group).
    static void keep_pool() {
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return;
        const unsigned long long bit = 1ull << dev;
        if (done.fetch_or(bit) & bit) return;
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            uint64_t keep = 6ull << 30;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};
// events of a timed run: destroyed on exit unless handed over to the plan (deferred profiling)
struct EventSet {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool keep = false;
    ~EventSet() {
        if (!keep)
            for (auto& e : ev)
                if (e) (void)hipEventDestroy(e);
    }
};

int ensure_device(icv_plan_t pl) {
    std::lock_guard<std::mutex> lk(pl->mu);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (pl->device == dev) return ICV_OK;
    if (pl->device >= 0) return fail(ICV_ERR_INVALID, "plan is bound to another device");
    const icv::Plan& p = pl->p;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    pl->n_cu = prop.multiProcessorCount;
    auto up = [&](const void* h, size_t bytes, void** d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    HIP_TRY(up(p.dst.data(), p.dst.size() * 4 + 0, (void**)&pl->d_dst));
    HIP_TRY(up(p.src.data(), p.src.size() * 4, (void**)&pl->d_src));
    HIP_TRY(up(p.w_start.data(), p.w_start.size() * 4, (void**)&pl->d_wstart));
    HIP_TRY(up(p.w_len.data(), p.w_len.size() * 4, (void**)&pl->d_wlen));
    HIP_TRY(up(p.w_denom.data(), p.w_denom.size() * 8, (void**)&pl->d_wdenom));
    HIP_TRY(up(p.pad_idx.data(), p.pad_idx.size() * 4, (void**)&pl->d_pad));
    HIP_TRY(up(p.w_pack.data(), p.w_pack.size() * 4, (void**)&pl->d_wpack));
    {
        const unsigned zeros[2] = {0u, 0u};
        HIP_TRY(up(zeros, sizeof(zeros), (void**)&pl->d_tie_n));
    }
    HIP_TRY(up(p.cov_col.data(), p.cov_col.size() * 4, (void**)&pl->d_cov_col));
    HIP_TRY(up(p.cov_j0.data(), p.cov_j0.size() * 4, (void**)&pl->d_cov_j0));
    HIP_TRY(up(p.cov_cnt.data(), p.cov_cnt.size() * 4, (void**)&pl->d_cov_cnt));
    {
        // tables of k_gene_fused (16-bit fields: the fused kernel applies where they fit)
        const size_t R = p.gv_run_j0.size();
        bool ok = p.W <= 65535 && R < 65535;
        std::vector<uint32_t> pk(R);
        for (size_t r = 0; r < R && ok; ++r) {
            ok = p.gv_run_cnt[r] <= 128 && p.gv_run_mult[r] <= 65535;  // (<= 128 windows per gene: gv_numpy_sum)
            pk[r] = (uint32_t)p.gv_run_j0[r] | ((uint32_t)p.gv_run_cnt[r] << 16);
        }
        std::vector<uint16_t> c16(((size_t)p.n_cols_all + 7) / 8 * 8, (uint16_t)R);
        for (int c = 0; c < p.n_cols_all && ok; ++c) c16[c] = p.gv_col_run[c] < 0 ? (uint16_t)R : (uint16_t)p.gv_col_run[c];
        pl->gv_fused_ok = ok;
        pl->gv_mult_bytes = 1;
        for (int32_t v : p.gv_run_mult)
            if (v > 255) pl->gv_mult_bytes = 2;
        HIP_TRY(up(pk.data(), pk.size() * 4, (void**)&pl->d_gv_pk));
        HIP_TRY(up(p.gv_run_mult.data(), p.gv_run_mult.size() * 4, (void**)&pl->d_gv_mult));
        HIP_TRY(up(c16.data(), c16.size() * 2, (void**)&pl->d_gv_col16));
    }
    HIP_TRY(up(p.dst16.data(), p.dst16.size() * 2, (void**)&pl->d_dst16));
    HIP_TRY(up(p.x16_wdesc.data(), p.x16_wdesc.size() * 4, (void**)&pl->d_x16_wdesc));
    HIP_TRY(up(p.blk_g0.data(), p.blk_g0.size() * 4, (void**)&pl->d_blk_g0));
    HIP_TRY(up(p.se_w0.data(), p.se_w0.size() * 4, (void**)&pl->d_se_w0));
    HIP_TRY(up(p.se_w1.data(), p.se_w1.size() * 4, (void**)&pl->d_se_w1));
    pl->zrow_elems = (size_t)icv::round_up(p.Gp, 4) + 4;  // >= Gp + 1: the trash slot reads 0
    HIP_TRY(hipMalloc(&pl->d_zrow, pl->zrow_elems * 8));
    pl->device = dev;
    return ICV_OK;
}

int check_matrix(const icv_plan_t pl, const icv_matrix* m) {
    if (!pl || !m) return fail(ICV_ERR_INVALID, "null plan or matrix");
    if (m->n_cols != pl->p.n_cols_all)
        return fail(ICV_ERR_INVALID, "matrix has " + std::to_string(m->n_cols) + " columns, plan expects " +
                                         std::to_string(pl->p.n_cols_all));
    if (m->dtype != ICV_F32 && m->dtype != ICV_F64) return fail(ICV_ERR_INVALID, "dtype must be ICV_F32 or ICV_F64");
    if (m->format != ICV_DENSE && m->format != ICV_CSR) return fail(ICV_ERR_INVALID, "format must be dense or csr");
    if (m->n_rows < 0) return fail(ICV_ERR_INVALID, "negative n_rows");
    if (m->n_rows > 0 && !m->values && !(m->format == ICV_CSR)) return fail(ICV_ERR_INVALID, "null values");
    if (m->format == ICV_DENSE && m->ld < m->n_cols) return fail(ICV_ERR_INVALID, "ld < n_cols");
    if (m->format == ICV_CSR && (!m->indptr || (!m->indices && m->values)))
        return fail(ICV_ERR_INVALID, "csr needs indptr and indices");
    return ICV_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Chromosome groups whose rows fit LDS at `elem_bytes` per gene (see icv_plan_s::parts).
int build_parts(icv_plan_t pl, int elem_bytes) {
    std::lock_guard<std::mutex> lk(pl->mu);
    if (pl->parts_elem >= elem_bytes) return ICV_OK;
    for (auto* q : pl->parts) icv_plan_destroy(q);
    pl->parts.clear();
    pl->part_woff.clear();
    pl->parts_elem = 0;
    const icv::Plan& p = pl->p;
    std::vector<int32_t> col_pos((size_t)p.n_cols_all, -1);
    {
        int rank = 0;
        for (int q = 0; q < p.Gp; ++q)
            if (p.src[q] >= 0) col_pos[p.src[q]] = rank++;
    }
    const int budget = (icv::kLdsLimit - icv::kScratchBytes - 64) / elem_bytes;  // padded genes per group
    int c0 = 0;
    while (c0 < p.n_chr) {
        int c1 = c0, genes = 0;
        while (c1 < p.n_chr) {
            const int g = p.chrom_off[c1 + 1] - p.chrom_off[c1];
            const int padded = (g + p.B - 1) / p.B * p.B;
            if (c1 > c0 && genes + padded > budget) break;
            genes += padded;
            ++c1;
        }
        const int lo = p.chrom_off[c0], hi = p.chrom_off[c1];
        std::vector<int32_t> cp((size_t)p.n_cols_all, -1), off((size_t)(c1 - c0 + 1));
        for (int g = 0; g < p.n_cols_all; ++g)
            if (col_pos[g] >= lo && col_pos[g] < hi) cp[g] = col_pos[g] - lo;
        for (int c = c0; c <= c1; ++c) off[c - c0] = p.chrom_off[c] - lo;
        icv_plan_s* part = new (std::nothrow) icv_plan_s();
        if (!part) return fail(ICV_ERR_NOMEM, "out of host memory");
        const std::string err =
            icv::build_plan(part->p, p.n_cols_all, cp.data(), c1 - c0, off.data(), p.window, p.step, p.B);
        if (!err.empty()) {
            delete part;
            return fail(ICV_ERR_INVALID, err);
        }
        const icv::Layout& pl_lay = elem_bytes == 4 ? part->p.lay32 : part->p.lay64;
        if (!pl_lay.fits) {
            icv_plan_destroy(part);
            for (auto* q : pl->parts) icv_plan_destroy(q);
            pl->parts.clear();
            pl->part_woff.clear();
            return fail(ICV_ERR_UNSUPPORTED,
                        "a single chromosome needs " + std::to_string(pl_lay.total) +
                            " bytes of LDS per workgroup (limit 163840): too many genes on one chromosome");
        }
        pl->parts.push_back(part);
        pl->part_woff.push_back(p.chr_pos[c0]);
        c0 = c1;
    }
    pl->parts_elem = elem_bytes;
    return ICV_OK;
}

int fill_params(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi, double lfc_clip,
                int32_t flags, float* out, int64_t ldo, double* cell_median, double* cell_stats, icv::KParams& K,
                const icv::Layout*& lay) {
    const icv::Plan& p = pl->p;
    if (!ref_lo) return fail(ICV_ERR_INVALID, "ref_lo is required");
    if (!(lfc_clip >= 0.0)) return fail(ICV_ERR_INVALID, "lfc_clip must be >= 0");
    if (!out || ldo < p.W) return fail(ICV_ERR_INVALID, "out is null or ldo < n_windows");
    lay = (m->dtype == ICV_F32) ? &p.lay32 : &p.lay64;
    if (!lay->fits) {  // chromosome-group fallback (smooth_split); fails if a single chromosome is too large
        const int rc = build_parts(pl, m->dtype == ICV_F32 ? 4 : 8);
        if (rc) return rc;
    }
    std::memset(&K, 0, sizeof(K));
    K.values = m->values;
    K.indptr = m->indptr;
    K.indices = m->indices;
    K.n_rows = m->n_rows;
    K.ld = m->ld;
    K.n_cols = m->n_cols;
    const int vn = (m->dtype == ICV_F32) ? 4 : 2;
    K.vec_ok = (m->format == ICV_DENSE) && aligned16(m->values) && aligned16(ref_lo) &&
               (!ref_hi || aligned16(ref_hi)) && (m->ld % vn == 0);
    K.ref_lo = ref_lo;
    K.ref_hi = ref_hi;
    K.zrow = pl->d_zrow;
    K.zrow_bytes = (int64_t)pl->zrow_elems * ((m->dtype == ICV_F32) ? 4 : 8);
    K.bounded = ref_hi != nullptr;
    K.trunc = flags & (ICV_FLAG_TRUNC_TO_INT | ICV_FLAG_ROUND_F32);
    K.cap = lfc_clip;
    K.dst = pl->d_dst;
    K.src = pl->d_src;
    K.w_start = pl->d_wstart;
    K.w_len = pl->d_wlen;
    K.w_denom = pl->d_wdenom;
    K.dst16 = pl->d_dst16;
    K.pad_idx = pl->d_pad;
    K.w_pack = pl->d_wpack;
    K.x16_wdesc = pl->d_x16_wdesc;
    K.blk_g0 = pl->d_blk_g0;
    K.x16_half = p.x16_half;
    K.n_pad = (int32_t)p.pad_idx.size();
    K.pyr_den = p.pyr_den;
    K.pyr_rcp = p.pyr_rcp;
    {
        const double capd = (m->dtype == ICV_F32) ? (double)(float)lfc_clip : lfc_clip;
        K.med_bound = capd * 1.000001 + 1e-30;
    }
    K.B = p.B;
    K.NB = p.NB;
    K.Gp = p.Gp;
    K.W = p.W;
    K.win_off = lay->win_off;
    K.scratch_off = lay->scratch_off;
    K.out = out;
    K.ldo = ldo;
    K.cell_median = cell_median;
    K.cell_stats = cell_stats;
    return ICV_OK;
}

int run_kernel(void (*kern)(const icv::KParams), int64_t grid, int lds, const icv::KParams& K, hipStream_t st,
               int block = icv::NT) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (knobs().phase_profile) {
        // developer diagnostic: shader cycles per phase, summed over workgroups (thread 0 of each)
        unsigned long long* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, 32 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(d, 0, 32 * sizeof(unsigned long long)));
        icv::KParams K2 = K;
        K2.dbg = d;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, st, K2);
        HIP_TRY(hipStreamSynchronize(st));
        unsigned long long h[32];
        HIP_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        if (block == icv::XT) {  // k_smooth_x16 (-DICV_X_PROFILE): shader cycles per cell of wavefront 15
            (void)hipFree(d);
            std::fprintf(stderr, "[icv x16 profile] grid=%lld rows=%lld lds=%d cycles per cell: A: store %.0f, chains %.0f, S "
                                 "%.0f, barrier 1 %.0f, B: output+gather+clear %.0f, W %.0f, L %.0f, barrier 2 %.0f\n", (long long)grid,
                         (long long)K.n_rows, lds, (double)h[6] / (double)K.n_rows, (double)h[7] / (double)K.n_rows,
                         (double)h[0] / (double)K.n_rows, (double)h[1] / (double)K.n_rows,
                         (double)h[4] / (double)K.n_rows, (double)h[5] / (double)K.n_rows,
                         (double)h[2] / (double)K.n_rows, (double)h[3] / (double)K.n_rows);
            return ICV_OK;
        }
        (void)hipFree(d);
        if (lds == icv::kSeLds) {  // k_smooth_se (-DICV_SE_PROFILE=<thread>): work of phase 0..3, each followed by its barrier wait
            std::fprintf(stderr, "[icv se profile] grid=%lld rows=%lld cycles per cell of the profiled thread:", (long long)grid,
                         (long long)K.n_rows);
            const char* nm[8] = {"ph0", "A", "ph1", "B1", "ph2", "B2", "ph3", "B3"};
            for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %s %.0f", nm[i], (double)h[i] / (double)K.n_rows);
            std::fprintf(stderr, "\n");
            return ICV_OK;
        }
        const char* names[6] = {"L load+scatter", "S block sums", "W windows", "M2 rank/select", "O output",
                                "M1 pivot search"};
        double tot = 0;
        for (int i = 0; i < 6; ++i) tot += (double)h[i];
        std::fprintf(stderr, "[icv phase profile] grid=%lld rows=%lld lds=%d\n", (long long)grid,
                     (long long)K.n_rows, lds);
        for (int i = 0; i < 6; ++i)
            std::fprintf(stderr, "  %-16s %12.0f cycles/cell  %5.1f %%\n", names[i], (double)h[i] / (double)K.n_rows,
                         100.0 * (double)h[i] / tot);
        std::fprintf(stderr, "  [6] %.2f  [7] %.4f per cell (fast: median iterations, split cells; ws: phases are "
                             "A-wait, gather+L, B1-wait, S+emit, S01+B3, W+B4; [6] = selector scan cycles, [7] = fallback cells)\n",
                     (double)h[6] / (double)K.n_rows, (double)h[7] / (double)K.n_rows);
        return ICV_OK;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, st, K);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

// the geometry / input admits k_smooth_x16 (same routing as launch_smooth -> launch_smooth_fast)
bool x16_applies(const icv_plan_t pl, const icv_matrix* m, const icv::KParams& K, const icv::Layout& lay) {
    const icv::Plan& p = pl->p;
    return lay.fits && m->dtype == ICV_F32 && m->format == ICV_DENSE && p.ws_ok && K.vec_ok && std::isfinite(K.cap) &&
           !knobs().force_generic && p.x16_ok && !K.bounded && !knobs().no_x16 &&
           p.step == 10 &&
           ((p.B == 10 && p.window == 100 && p.x16_fine == 4096) || (p.B == 5 && p.window == 250 && p.x16_fine == 1024));
}

// plan-owned workspace of the kernels that hand cells back to the generic k_smooth: the list of those cells, its
// counter (zeroed here) and the per-wavefront partial moments
int hand_back_workspace(icv_plan_t pl, icv::KParams& K, hipStream_t st) {
    if (pl->row_list_cap < K.n_rows) {
        (void)hipFree(pl->d_row_list);
        pl->d_row_list = nullptr;
        pl->row_list_cap = 0;
        HIP_TRY(hipMalloc((void**)&pl->d_row_list, (size_t)K.n_rows * sizeof(int64_t)));
        pl->row_list_cap = K.n_rows;
    }
    if (pl->cell_part_cap < K.n_rows) {
        (void)hipFree(pl->d_cell_part);
        pl->d_cell_part = nullptr;
        pl->cell_part_cap = 0;
        // 16 wavefront partial pairs per cell (k_smooth_x16; k_smooth_ws / k_smooth_se use the first 8)
        HIP_TRY(hipMalloc((void**)&pl->d_cell_part, (size_t)K.n_rows * 32 * sizeof(double)));
        pl->cell_part_cap = K.n_rows;
    }
    K.cell_part = pl->d_cell_part;
    if (!pl->d_row_count) HIP_TRY(hipMalloc((void**)&pl->d_row_count, sizeof(int)));
    HIP_TRY(hipMemsetAsync(pl->d_row_count, 0, sizeof(int), st));
    K.row_list = pl->d_row_list;
    K.row_count = pl->d_row_count;
    return ICV_OK;
}

// cells a fast kernel handed back (more than 64 windows in the median bins, a stored NaN): the generic kernel
// recomputes them (it reads the device-side count and exits at once when the list is empty)
int launch_hand_back(icv_plan_t pl, const icv::KParams& K, hipStream_t st, bool csr) {
    const icv::Plan& p = pl->p;
    icv::KParams G = K;
    G.win_off = p.lay32.win_off;
    G.scratch_off = p.lay32.scratch_off;
    G.dbg = nullptr;
    const int need = (p.NB + icv::kThreads - 1) / icv::kThreads;
    void (*gk)(const icv::KParams);
    if (csr) gk = need <= 4 ? icv::k_smooth<float, true, 4> : icv::k_smooth<float, true, 8>;
    else gk = need <= 4 ? icv::k_smooth<float, false, 4> : icv::k_smooth<float, false, 8>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                p.lay32.total));
    int64_t g2 = pl->n_cu;
    if (g2 > K.n_rows) g2 = K.n_rows;
    hipLaunchKernelGGL(gk, dim3((unsigned)g2), dim3(icv::NT), p.lay32.total, st, G);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

// k_smooth_se: fraction bits of its bins.  A bin starts at 1.5 * 2^52 and must stay within 2^51 of it: at most B
// entries of |d| <= 2 cap each in S0 (units 2^-k0), sum of j |d| <= B (B - 1) / 2 * 2 cap in S1 (units 2^-k1).  Fewer
// than 40 bits (a clip value beyond ~100) and the input takes the kernels that build the row in LDS.
bool se_fraction_bits(const icv::Plan& p, double cap, int* k0, int* k1) {
    if (!p.se_ok || knobs().no_sd) return false;  // ICV_NO_SD: developer knob, the kernels with a row in LDS
    int e = 0;
    (void)std::frexp((double)p.B * 2.0 * cap + 1.0, &e);  // < 2^e
    const int a = 51 - e;
    (void)std::frexp((double)p.B * (double)(p.B - 1) * cap + 1.0, &e);
    const int b = 51 - e;
    if (a < 40 || b < 36) return false;
    *k0 = a;
    *k1 = b;
    return true;
}

// the CSR float32 input takes the stored-entries kernel k_smooth_se (same test in launch_smooth and in the chunk-moment
// set-up of icv_infercnv_run)
bool stored_entries_kernel(const icv_plan_t pl, const icv_matrix* m, const icv::KParams& K, const icv::Layout& lay) {
    if (!(lay.fits && m->dtype == ICV_F32 && m->format == ICV_CSR && std::isfinite(K.cap) &&
          m->csr_end > m->csr_begin && !knobs().force_generic))
        return false;
    int k0 = 0, k1 = 0;
    return se_fraction_bits(pl->p, K.cap, &k0, &k1);
}

// k_smooth_se (icv_kernel_se.hpp): CSR float32 input in block form, stored entries only
int launch_smooth_se(icv_plan_t pl, icv::KParams K, hipStream_t st, hipEvent_t kernel_done = nullptr,
                     int64_t csr_entries = 0, int row_len_hint = 0) {
    const icv::Plan& p = pl->p;
    int k0 = 0, k1 = 0;
    if (!se_fraction_bits(p, K.cap, &k0, &k1)) return fail(ICV_ERR_INVALID, "k_smooth_se does not apply");
    AsyncBuf tab_guard, hi_guard, wt_guard, g_guard;  // per-column, per-window, per-block tables: released on every exit path
    const int nz = (int)pl->zrow_elems;
    hipLaunchKernelGGL(icv::k_zero_row<float>, dim3((nz + 255) / 256), dim3(256), 0, st, K,
                       static_cast<float*>(pl->d_zrow), nz);
    HIP_TRY(tab_guard.alloc((size_t)K.n_cols * 16, st));
    HIP_TRY(wt_guard.alloc((size_t)p.W * 16, st));
    HIP_TRY(g_guard.alloc((size_t)(p.NB + 8) * 8, st));
    if (K.bounded) HIP_TRY(hi_guard.alloc((size_t)K.n_cols * 4, st));
    K.sd_tab = tab_guard.p;
    K.sd_tab_hi = K.bounded ? hi_guard.as<float>() : nullptr;
    K.sd_wtab = wt_guard.p;
    K.sd_g16 = g_guard.as<double>();
    K.sd_scale = std::ldexp(1.0, k0);
    K.sd_qinv = std::ldexp(1.0, -k0);
    K.sd_q1inv = std::ldexp(1.0, -k1);
    K.sd_r = std::ldexp(1.0, k1 - k0);
    K.sd_window = p.window;
    hipLaunchKernelGGL(icv::k_se_table, dim3((unsigned)((K.n_cols + 255) / 256)), dim3(256), 0, st, K,
                       tab_guard.as<icv::u32x4>(), const_cast<float*>(K.sd_tab_hi), (float)std::ldexp(1.0, k1));
    const int n_wt = p.W > p.NB + 8 ? p.W : p.NB + 8;
    hipLaunchKernelGGL(icv::k_se_wtab, dim3((unsigned)((n_wt + 255) / 256)), dim3(256), 0, st, K,
                       static_cast<const float*>(pl->d_zrow), pl->d_se_w0, pl->d_se_w1, wt_guard.as<icv::u32x4>(),
                       g_guard.as<double>(), K.sd_r);
    if (int rc = hand_back_workspace(pl, K, st)) return rc;
    int64_t grid = se_grid(pl, K.n_rows);
    if (grid < 1) {
        if (kernel_done) HIP_TRY(hipEventRecord(kernel_done, st));
        return ICV_OK;
    }
    void (*kern)(const icv::KParams);
    // window registers per thread: 3 where the windows fit (<= 1 536: window 250 / step 10 at 20 000 genes has 1 472), else
    // 4; entry slots per thread from the mean row length: PF x 512 slots should hold mean + 4 sigma (binomial) entries --
    // longer rows take the in-phase loop, any PF gives the same bits
    const bool w3 = p.W <= 3 * icv::NT && !knobs().se_maxw4;
    int pf = 4;
    if (K.chunk_part && !knobs().se_maxw4 && K.n_rows > 0) {
        double want = 0.0;
        if (row_len_hint > 0) {
            want = (double)row_len_hint;  // the caller's figure: the length most rows stay under (icv_matrix._pad)
        } else if (csr_entries > 0) {     // none: mean + 3.5 sigma of a binomial row length
            const double mean = (double)csr_entries / (double)K.n_rows;
            const double dens = mean / (double)(K.n_cols > 0 ? K.n_cols : 1);
            want = mean + 3.5 * std::sqrt(mean * (dens < 1.0 ? 1.0 - dens : 0.0)) + 8.0;
        }
        if (want > 0.0) pf = want <= icv::NT ? 1 : want <= 2 * icv::NT ? 2 : want <= 3 * icv::NT ? 3 : 4;
    }
#define ICV_SE_PICK(MW, CH, BD)                                                                              \
    (pf == 1 ? icv::k_smooth_se<MW, CH, BD, 1> : pf == 2 ? icv::k_smooth_se<MW, CH, BD, 2>                   \
     : pf == 3 ? icv::k_smooth_se<MW, CH, BD, 3> : icv::k_smooth_se<MW, CH, BD, 4>)
    if (K.win_out) {
        // calculate_gene_values: the float64 windows leave the kernel as well (four entry slots per thread whatever
        // the row length: any slot count gives the same bits, and this path is bound by the cells x genes output)
        if (!K.chunk_part) return fail(ICV_ERR_INVALID, "k_smooth_se with windows needs the chunk-moment mode");
        if (w3) kern = K.bounded ? icv::k_smooth_se<3, true, true, 4, true> : icv::k_smooth_se<3, true, false, 4, true>;
        else kern = K.bounded ? icv::k_smooth_se<4, true, true, 4, true> : icv::k_smooth_se<4, true, false, 4, true>;
    } else if (K.chunk_part) {
        if (w3) kern = K.bounded ? ICV_SE_PICK(3, true, true) : ICV_SE_PICK(3, true, false);
        else kern = K.bounded ? ICV_SE_PICK(4, true, true) : ICV_SE_PICK(4, true, false);
    } else {
        if (w3) kern = K.bounded ? icv::k_smooth_se<3, false, true> : icv::k_smooth_se<3, false, false>;
        else kern = K.bounded ? icv::k_smooth_se<4, false, true> : icv::k_smooth_se<4, false, false>;
    }
#undef ICV_SE_PICK
    int rc = run_kernel(kern, grid, icv::kSeLds, K, st);
    if (kernel_done && !rc) HIP_TRY(hipEventRecord(kernel_done, st));
    if (rc) return rc;
    if (!K.chunk_part)
        hipLaunchKernelGGL(icv::k_stats_finish, dim3((unsigned)((K.n_rows + 255) / 256)), dim3(256), 0, st, K.cell_part,
                           K.n_rows, K.cell_stats);
    pl->last_kernel = ICV_KERNEL_SD;
    return launch_hand_back(pl, K, st, true);
}

// float32, blocked form, small enough geometry: register-prefetch kernels (dense or prepared CSR)
// kernel_done (optional): recorded right after the smoothing kernel itself, before the moment finish / hand-back
// launches, so that profiling reports the dominant kernel's own duration
int launch_smooth_fast(icv_plan_t pl, icv::KParams K, hipStream_t st, bool csr, int64_t csr_begin, int64_t csr_end,
                       hipEvent_t kernel_done = nullptr) {
    const icv::Plan& p = pl->p;
    const int need_b = (p.NB + icv::kThreads - 1) / icv::kThreads;
    void (*kern)(const icv::KParams) = nullptr;
    constexpr int U = icv::kFastUMax;
    K.scratch_off = p.fast_scratch_off;
    AsyncBuf ws_guard;  // prepared CSR entries: released on every exit path
    void* ws_buf = nullptr;
    const int lds = p.fast_lds;
    if (!p.ws_ok) return -1;  // caller falls back to the generic kernel
    {
        const bool u10 = (p.B == 10 && p.window == 100);
        if (!csr) {
            if (need_b <= 4) kern = u10 ? icv::k_smooth_ws<U, 4, 4, 10, 10, false> : icv::k_smooth_ws<U, 4, 4, 0, 0, false>;
            else if (p.B == 5 && p.window == 250) kern = icv::k_smooth_ws<U, 8, 4, 5, 50, false>;
            else kern = (p.B == 5) ? icv::k_smooth_ws<U, 8, 4, 5, 0, false> : icv::k_smooth_ws<U, 8, 4, 0, 0, false>;
        } else {
            const int nz = (int)pl->zrow_elems;
            hipLaunchKernelGGL(icv::k_zero_row<float>, dim3((nz + 255) / 256), dim3(256), 0, st, K,
                               static_cast<float*>(pl->d_zrow), nz);
            const int64_t n = csr_end - csr_begin;
            int64_t g = (n + 255) / 256;
            if (g > 8192) g = 8192;
            if (need_b <= 4) kern = u10 ? icv::k_smooth_ws<U, 4, 4, 10, 10, true> : icv::k_smooth_ws<U, 4, 4, 0, 0, true>;
            else kern = (p.B == 5) ? icv::k_smooth_ws<U, 8, 4, 5, 0, true> : icv::k_smooth_ws<U, 8, 4, 0, 0, true>;
            // prepared entries {LDS position, centred and clipped value} on top of the zero row
            HIP_TRY(ws_guard.alloc((size_t)(n > 0 ? n : 1) * 6, st));
            ws_buf = ws_guard.p;
            float* cv = static_cast<float*>(ws_buf);
            uint16_t* ps = reinterpret_cast<uint16_t*>(cv + (n > 0 ? n : 1));
            K.cvals = cv - csr_begin;  // indexed by the absolute entry number
            K.pos16 = ps - csr_begin;
            if (n > 0)
                hipLaunchKernelGGL(icv::k_csr_prepare, dim3((unsigned)g), dim3(256), 0, st, K, csr_begin, csr_end,
                                   const_cast<uint16_t*>(K.pos16), const_cast<float*>(K.cvals));
        }
        K.hist_off = p.ws_hist_off;
        if (int rc = hand_back_workspace(pl, K, st)) return rc;
    }
    int per_cu = icv::kLdsLimit / lds;
    if (per_cu > 4) per_cu = 4;
    if (const int v = knobs().wgs_per_cu)  // developer knob: occupancy experiments
        if (v >= 1 && v < per_cu) per_cu = v;
    int64_t grid = (int64_t)pl->n_cu * per_cu;
    if (grid > K.n_rows) grid = K.n_rows;
    if (grid < 1) {
        if (kernel_done) HIP_TRY(hipEventRecord(kernel_done, st));
        return ICV_OK;
    }
    int rc;
    // dense float32, one reference row, window 100 / step 10 or window 250 / step 10 geometry: the 16-wavefront
    // kernel, one 1024-thread workgroup per CU (ICV_NO_X16=1: developer knob, previous generation)
    void (*xk)(const icv::KParams) = nullptr;
    if (!csr && p.x16_ok && !K.bounded && !knobs().no_x16) {
        if (p.step != 10)
            xk = nullptr;  // the instantiations below assume step 10 (blocks between adjacent windows)
        else if (p.B == 10 && p.window == 100 && p.x16_fine == 4096)
            xk = K.win_out ? icv::k_smooth_x16<10, 10, 1, true, 4096, true, true>
                           : (K.chunk_part ? icv::k_smooth_x16<10, 10, 1, true, 4096, true>
                                           : icv::k_smooth_x16<10, 10, 1, false, 4096, true>);
        else if (p.B == 5 && p.window == 250 && p.x16_fine == 1024)
            xk = K.win_out ? icv::k_smooth_x16<5, 50, 2, true, 1024, false, true>
                           : (K.chunk_part ? icv::k_smooth_x16<5, 50, 2, true, 1024, false>
                                           : icv::k_smooth_x16<5, 50, 2, false, 1024, false>);
    }
    // float64 windows requested (calculate_gene_values): only the x16 instantiations write them, in chunk-moment mode;
    // every other geometry takes the generic kernel (ONE smoothing pass either way)
    if (K.win_out && (!xk || !K.chunk_part)) return -1;
    if (xk) {
        icv::KParams X = K;
        X.win_off = p.x16_s01_off;
        X.hist_off = p.x16_hist_off;
        X.scratch_off = p.x16_scratch_off;
        int64_t gx = pl->n_cu;
        if (gx > K.n_rows) gx = K.n_rows;
        rc = run_kernel(xk, gx, p.x16_lds, X, st, icv::XT);
        if (kernel_done && !rc) HIP_TRY(hipEventRecord(kernel_done, st));
        if (rc) return rc;
        if (!K.chunk_part)
            hipLaunchKernelGGL(icv::k_stats_finish_n, dim3((unsigned)((K.n_rows + 255) / 256)), dim3(256), 0, st,
                               K.cell_part, K.n_rows, icv::XWAVE, K.cell_stats);
    } else {
        rc = run_kernel(kern, grid, lds, K, st);
        if (kernel_done && !rc) HIP_TRY(hipEventRecord(kernel_done, st));
        if (rc) return rc;
        // per-wavefront partial moments -> cell_stats (cells handed back are overwritten by k_smooth below)
        hipLaunchKernelGGL(icv::k_stats_finish, dim3((unsigned)((K.n_rows + 255) / 256)), dim3(256), 0, st,
                           K.cell_part, K.n_rows, K.cell_stats);
    }
    pl->last_kernel = xk ? ICV_KERNEL_X16 : (csr ? ICV_KERNEL_WS_CSR : ICV_KERNEL_WS);
    return launch_hand_back(pl, K, st, csr);
}

template <typename T, bool CSR>
int launch_smooth_t(icv_plan_t pl, const icv::KParams& K, const icv::Layout& lay, hipStream_t st) {
    const icv::Plan& p = pl->p;
    int maxb = 0;
    if (p.B > 1) {
        const int need = (p.NB + icv::kThreads - 1) / icv::kThreads;
        maxb = need <= 4 ? 4 : 8;
    }
    void (*kern)(const icv::KParams) = nullptr;
    if (maxb == 0) kern = icv::k_smooth<T, CSR, 0>;
    else if (maxb == 4) kern = icv::k_smooth<T, CSR, 4>;
    else kern = icv::k_smooth<T, CSR, 8>;
    int per_cu = icv::kLdsLimit / lay.total;
    if (per_cu > 4) per_cu = 4;  // 32 wavefronts per CU / 8 per workgroup
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (int64_t)pl->n_cu * per_cu;
    if (grid > K.n_rows) grid = K.n_rows;
    if (grid < 1) return ICV_OK;
    if constexpr (CSR) {
        const int n = (int)pl->zrow_elems;
        hipLaunchKernelGGL(icv::k_zero_row<T>, dim3((n + 255) / 256), dim3(256), 0, st, K,
                           static_cast<T*>(pl->d_zrow), n);
    }
    if (lay.win_global) {
        const int64_t need = grid * (int64_t)p.W;
        if (pl->win_scratch_cap < need) {
            (void)hipFree(pl->d_win_scratch);
            pl->d_win_scratch = nullptr;
            HIP_TRY(hipMalloc((void**)&pl->d_win_scratch, (size_t)need * sizeof(double)));
            pl->win_scratch_cap = need;
        }
        icv::KParams K2 = K;
        K2.win_scratch = pl->d_win_scratch;
        return run_kernel(kern, grid, lay.total, K2, st);
    }
    return run_kernel(kern, grid, lay.total, K, st);
}

// float64 windows of all cells, chromosome group by chromosome group (every pass re-reads the rows and keeps only
// its group's genes), into win[n_rows x W].  K: parameters of the full plan (fill_params).
int split_windows(icv_plan_t pl, const icv_matrix* m, const icv::KParams& K, double* win, hipStream_t st) {
    const bool f32 = m->dtype == ICV_F32;
    for (size_t k = 0; k < pl->parts.size(); ++k) {
        icv_plan_t part = pl->parts[k];
        int rc = ensure_device(part);
        if (rc) return rc;
        const icv::Plan& q = part->p;
        const icv::Layout& lay = f32 ? q.lay32 : q.lay64;
        icv::KParams P = K;
        P.dst = part->d_dst;
        P.src = part->d_src;
        P.w_start = part->d_wstart;
        P.w_len = part->d_wlen;
        P.w_denom = part->d_wdenom;
        P.dst16 = part->d_dst16;
        P.pad_idx = part->d_pad;
        P.w_pack = part->d_wpack;
        P.n_pad = (int32_t)q.pad_idx.size();
        P.B = q.B;
        P.NB = q.NB;
        P.Gp = q.Gp;
        P.W = q.W;
        P.win_off = lay.win_off;
        P.scratch_off = lay.scratch_off;
        P.zrow = part->d_zrow;
        P.zrow_bytes = (int64_t)part->zrow_elems * (f32 ? 4 : 8);
        P.win_out = win + pl->part_woff[k];
        P.win_ld = pl->p.W;
        P.win_only = 1;
        P.row_list = nullptr;
        P.row_count = nullptr;
        if (f32)
            rc = m->format == ICV_DENSE ? launch_smooth_t<float, false>(part, P, lay, st)
                                        : launch_smooth_t<float, true>(part, P, lay, st);
        else
            rc = m->format == ICV_DENSE ? launch_smooth_t<double, false>(part, P, lay, st)
                                        : launch_smooth_t<double, true>(part, P, lay, st);
        if (rc) return rc;
    }
    return ICV_OK;
}

// steps 1-4 for a gene set whose row does not fit LDS: windows by chromosome group, then median and centring
// on the float64 windows in HBM
int smooth_split(icv_plan_t pl, const icv_matrix* m, const icv::KParams& K, hipStream_t st) {
    const int64_t n = K.n_rows;
    if (n < 1) return ICV_OK;
    const int W = pl->p.W;
    AsyncBuf win_b, med_b;
    // (a caller that wants the float64 windows -- calculate_gene_values -- with a contiguous buffer gets them here)
    const bool own = !(K.win_out && K.win_ld == W);
    if (own) HIP_TRY(win_b.alloc((size_t)n * W * sizeof(double), st));
    HIP_TRY(med_b.alloc((size_t)n * sizeof(double), st));
    double *win = own ? win_b.as<double>() : K.win_out, *med = med_b.as<double>();
    int rc = split_windows(pl, m, K, win, st);
    if (!rc && own && K.win_out)
        HIP_TRY(hipMemcpy2DAsync(K.win_out, (size_t)K.win_ld * sizeof(double), win, (size_t)W * sizeof(double),
                                 (size_t)W * sizeof(double), (size_t)n, hipMemcpyDeviceToDevice, st));
    if (!rc) {
        hipLaunchKernelGGL(icv::k_row_median, dim3((unsigned)n), dim3(256), 0, st, win, n, W, med);
        hipLaunchKernelGGL(icv::k_win_finish, dim3((unsigned)n), dim3(256), 0, st, win, n, W, med, K.out, K.ldo,
                           K.cell_median, K.cell_stats);
        if (hipGetLastError() != hipSuccess) rc = fail(ICV_ERR_HIP, "split smoothing launch failed");
    }
    return rc;
}

int launch_smooth(icv_plan_t pl, const icv_matrix* m, const icv::KParams& K, const icv::Layout& lay,
                  hipStream_t st, hipEvent_t kernel_done = nullptr, bool* recorded = nullptr) {
    if (recorded) *recorded = false;
    if (!lay.fits) {
        pl->last_kernel = ICV_KERNEL_SPLIT;
        return smooth_split(pl, m, K, st);
    }
    const bool want_win = K.win_out != nullptr;
    const bool fast_allowed = m->dtype == ICV_F32 && std::isfinite(K.cap) && !knobs().force_generic;
    if (fast_allowed && m->format == ICV_DENSE && pl->p.ws_ok && K.vec_ok) {
        const int rc = launch_smooth_fast(pl, K, st, false, 0, 0, kernel_done);
        if (rc >= 0) {
            if (recorded) *recorded = kernel_done != nullptr;
            return rc;
        }
    }
    if (fast_allowed && m->format == ICV_CSR && m->csr_end > m->csr_begin) {
        if (stored_entries_kernel(pl, m, K, lay) && (!want_win || K.chunk_part)) {
            if (recorded) *recorded = kernel_done != nullptr;
            return launch_smooth_se(pl, K, st, kernel_done, m->csr_end - m->csr_begin, m->_pad > 0 ? m->_pad : 0);
        }
        if (pl->p.ws_ok && aligned16(K.ref_lo) && !want_win) {
            const int rc = launch_smooth_fast(pl, K, st, true, m->csr_begin, m->csr_end, kernel_done);
            if (rc >= 0) {
                if (recorded) *recorded = kernel_done != nullptr;
                return rc;
            }
        }
    }
    pl->last_kernel = ICV_KERNEL_GENERIC;
    if (m->dtype == ICV_F32)
        return m->format == ICV_DENSE ? launch_smooth_t<float, false>(pl, K, lay, st)
                                      : launch_smooth_t<float, true>(pl, K, lay, st);
    return m->format == ICV_DENSE ? launch_smooth_t<double, false>(pl, K, lay, st)
                                  : launch_smooth_t<double, true>(pl, K, lay, st);
}

int launch_apply(const icv_matrix* m, const icv::KParams& K, const double* thr, int64_t chunksize,
                 int64_t row_phase, hipStream_t st) {
    if (K.n_rows < 1) return ICV_OK;
    dim3 grid((unsigned)K.n_rows), block(256);
    if (m->dtype == ICV_F32) {
        if (m->format == ICV_DENSE)
            hipLaunchKernelGGL((icv::k_apply_thr<float, false>), grid, block, 0, st, K, thr, chunksize, row_phase);
        else
            hipLaunchKernelGGL((icv::k_apply_thr<float, true>), grid, block, 0, st, K, thr, chunksize, row_phase);
    } else {
        if (m->format == ICV_DENSE)
            hipLaunchKernelGGL((icv::k_apply_thr<double, false>), grid, block, 0, st, K, thr, chunksize, row_phase);
        else
            hipLaunchKernelGGL((icv::k_apply_thr<double, true>), grid, block, 0, st, K, thr, chunksize, row_phase);
    }
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

}  // namespace

// ---- reference profile in the reference's evaluation order (csrc/icv_kernel_chain.hpp) ------------------------------
namespace {
int current_cu_count() {  // (hipGetDeviceProperties costs a fraction of a millisecond: asked once per device)
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

// line -> tile of ChainLaunch's split (tile t owns the lines [t * n_lines / grid, (t + 1) * n_lines / grid))
__global__ void k_chain_line_tiles(int n_lines, int grid, uint16_t* line_tile) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= grid) return;
    const int l0 = (int)((int64_t)t * n_lines / grid), l1 = (int)((int64_t)(t + 1) * n_lines / grid);
    for (int l = l0; l < l1; ++l) line_tile[l] = (uint16_t)t;
}

template <typename T>
__global__ void k_chain_mean(const T* acc, int n, T denom, T* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = acc[i] / denom;  // IEEE division (numpy: true_divide(sum, n) in the matrix dtype)
}

template <typename T>
int colchain_dense(const icv_matrix* m, const int32_t* rows, int64_t n_sel, T* acc, hipStream_t st) {
    const icv::ChainLaunch L(m->n_cols, (int)sizeof(T), current_cu_count());
    constexpr int EPL = 16 / (int)sizeof(T);
    // the buffer's last row goes through the guarded tail when a 16-byte segment load could run past the end; with a
    // row list the kernel itself looks whether the list ends with that row (one scalar load: nothing is read back)
    int64_t tail = -1, n_dma = n_sel;
    if ((int64_t)(m->_pad > 0 ? m->_pad : 0) + (int64_t)(m->n_cols + EPL - 1) / EPL * EPL > m->ld) {
        tail = m->n_rows - 1;
        if (!rows) n_dma = n_sel - 1;
    }
    typedef void (*kern_t)(const T*, int64_t, int, int, int, const int32_t*, int64_t, int64_t, T*);
    const kern_t kern = rows ? (kern_t)icv::k_colchain<T, true> : (kern_t)icv::k_colchain<T, false>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                icv::kChLdsFull));
    hipLaunchKernelGGL(kern, dim3(L.grid), dim3(icv::kChThreads), L.lds_bytes, st, (const T*)m->values, m->ld,
                       m->n_cols, L.n_lines, L.lds_bytes, rows, n_dma, tail, acc);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

template <typename T>
int colchain_csr(const icv_matrix* m, const int32_t* rows, int64_t n_sel, double scale, T* acc, hipStream_t st) {
    const icv::ChainLaunch L(m->n_cols, (int)sizeof(T), current_cu_count());
    if (L.grid > 65535) return fail(ICV_ERR_UNSUPPORTED, "icv_colchain: more than 65535 column tiles");
    AsyncBuf lt_b, bounds_b;
    HIP_TRY(lt_b.alloc((size_t)L.n_lines * sizeof(uint16_t), st));
    const int esz_shift_q = sizeof(T) == 4 ? 2 : 3;
    const size_t tab_lds = (size_t)icv::kQTabRows * (size_t)((L.grid + 2) | 1) * sizeof(uint16_t) +
                           (size_t)L.n_lines * sizeof(uint16_t);
    if (m->n_cols <= 65535 && tab_lds <= 160 * 1024 && !knobs().no_chain_queues) {
        // per-column queues (csrc/icv_kernel_chainq.hpp): 16-bit tile-major bounds table + the streamed chain
        hipLaunchKernelGGL(k_chain_line_tiles, dim3((L.grid + 255) / 256), dim3(256), 0, st, L.n_lines, L.grid,
                           lt_b.as<uint16_t>());
        const int64_t n_tb = (n_sel + icv::kQTabRows - 1) / icv::kQTabRows;
        HIP_TRY(bounds_b.alloc((size_t)n_tb * (L.grid + 1) * icv::kQTabRows * sizeof(uint16_t), st));
        typedef void (*tb_t)(const int64_t*, const int32_t*, const int32_t*, int64_t, const uint16_t*, int, int, int,
                             uint32_t*);
        const tb_t tb = rows ? (tb_t)icv::k_csr_tile_bounds16<true> : (tb_t)icv::k_csr_tile_bounds16<false>;
        if (tab_lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tb), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tab_lds));
        hipLaunchKernelGGL(tb, dim3((unsigned)n_tb), dim3(256), tab_lds, st, m->indptr, m->indices, rows, n_sel,
                           lt_b.as<uint16_t>(), L.n_lines, esz_shift_q, L.grid, bounds_b.as<uint32_t>());
        typedef void (*kq_t)(const T*, const int64_t*, const int32_t*, int64_t, const int32_t*, int64_t, int, int, int,
                             const uint16_t*, T, int, T*);
        // lanes per row (= entry slots / 4) from the mean number of entries a row has in the WIDEST tile (from the
        // caller's row-length hint, icv_matrix._pad: the length most rows stay under, where there is one).  Measured on
        // 500 000 x 20 000 (profiles/r06_csr_means_density.txt): 16 slots up to ~8 entries per row and tile (beyond, too
        // many rounds hold a row past the slots, whose guarded loads wait behind the loads in flight), 32 slots to ~28.
        // Rows beyond the slots take the guarded loads: any choice gives the same bits.
        int pieces = 4;
        {
            const double rows_all = m->n_rows > 0 ? (double)m->n_rows : 1.0;
            double per_row = (double)(m->csr_end - m->csr_begin) / rows_all;
            if (m->_pad > 0 && (double)m->_pad > per_row) per_row = (double)m->_pad;
            const int tile_lines = (L.n_lines + L.grid - 1) / L.grid;
            const double mu = per_row * (double)(tile_lines * (128 / (int)sizeof(T))) / (double)(m->n_cols > 0 ? m->n_cols : 1);
            pieces = mu <= 2.0 ? 2 : mu <= 8.5 ? 4 : mu <= 28.0 ? 8 : 16;
            const int forced = knobs().chain_pieces;
            if (forced == 2 || forced == 4 || forced == 8 || forced == 16) pieces = forced;
        }
#define ICV_KQ(P) (rows ? (kq_t)icv::k_colchain_csrq<T, true, P> : (kq_t)icv::k_colchain_csrq<T, false, P>)
        const kq_t kq = pieces == 2 ? ICV_KQ(2) : pieces == 4 ? ICV_KQ(4) : pieces == 8 ? ICV_KQ(8) : ICV_KQ(16);
#undef ICV_KQ
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    icv::kChLdsFull));
        // (always the whole LDS of a CU: the ring of LDS rows is the work in flight; with more tiles than CUs the
        // workgroups take the CUs in turns)
        hipLaunchKernelGGL(kq, dim3(L.grid), dim3(icv::kChThreads), icv::kChLdsFull, st, (const T*)m->values, m->indptr,
                           m->indices, m->n_rows, rows, n_sel, m->n_cols, L.n_lines, icv::kChLdsFull,
                           (const uint16_t*)bounds_b.as<uint16_t>(), (T)scale,
                           knobs().chain_far > 0 ? knobs().chain_far : icv::kQFar, acc);
        HIP_TRY(hipGetLastError());
        return ICV_OK;
    }
    const int64_t n_blk = (n_sel + icv::kCcBlock - 1) / icv::kCcBlock;
    HIP_TRY(bounds_b.alloc((size_t)n_blk * (L.grid + 1) * icv::kCcBlock * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_chain_line_tiles, dim3((L.grid + 255) / 256), dim3(256), 0, st, L.n_lines, L.grid,
                       lt_b.as<uint16_t>());
    const int esz_shift = sizeof(T) == 4 ? 2 : 3;
    if (rows)
        hipLaunchKernelGGL(icv::k_csr_tile_bounds<true>, dim3((unsigned)n_blk), dim3(256), 0, st, m->indptr, m->indices,
                           rows, n_sel, lt_b.as<uint16_t>(), esz_shift, L.grid, bounds_b.as<uint32_t>());
    else
        hipLaunchKernelGGL(icv::k_csr_tile_bounds<false>, dim3((unsigned)n_blk), dim3(256), 0, st, m->indptr, m->indices,
                           rows, n_sel, lt_b.as<uint16_t>(), esz_shift, L.grid, bounds_b.as<uint32_t>());
    typedef void (*kern_t)(const T*, const int64_t*, const int32_t*, int64_t, const int32_t*, int64_t, int, int, int,
                           const uint32_t*, T, T*);
    const kern_t kern = rows ? (kern_t)icv::k_colchain_csr<T, true> : (kern_t)icv::k_colchain_csr<T, false>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                icv::kChLdsFull));
    hipLaunchKernelGGL(kern, dim3(L.grid), dim3(icv::kChThreads), L.lds_bytes, st, (const T*)m->values, m->indptr,
                       m->indices, m->n_rows, rows, n_sel, m->n_cols, L.n_lines, L.lds_bytes,
                       (const uint32_t*)bounds_b.as<uint32_t>(), (T)scale, acc);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}
}  // namespace


extern "C" {

const char* icv_last_error(void) { return g_err.c_str(); }
int icv_version(void) { return 100; }

void icv_developer_knobs_reload(void) {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    g_knobs.load();
    g_knobs_loaded = true;
}

int icv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int icv_plan_create(int32_t n_cols_all, const int32_t* h_col_pos, int32_t n_chr, const int32_t* h_chrom_offsets,
                    int32_t window, int32_t step, icv_plan_t* out) {
    if (!h_col_pos || !h_chrom_offsets || !out) return fail(ICV_ERR_INVALID, "null argument");
    icv_plan_s* pl = new (std::nothrow) icv_plan_s();
    if (!pl) return fail(ICV_ERR_NOMEM, "out of host memory");
    std::string err;
    try {
        err = icv::build_plan(pl->p, n_cols_all, h_col_pos, n_chr, h_chrom_offsets, window, step);
    } catch (const std::bad_alloc&) {
        delete pl;
        return fail(ICV_ERR_NOMEM, "out of host memory");
    }
    if (!err.empty()) {
        delete pl;
        return fail(ICV_ERR_INVALID, err);
    }
    *out = pl;
    return ICV_OK;
}

void icv_plan_destroy(icv_plan_t pl) {
    if (!pl) return;
    for (auto* q : pl->parts) icv_plan_destroy(q);
    pl->parts.clear();
    if (pl->device >= 0) {
        (void)hipFree(pl->d_dst);
        (void)hipFree(pl->d_src);
        (void)hipFree(pl->d_wstart);
        (void)hipFree(pl->d_wlen);
        (void)hipFree(pl->d_wdenom);
        (void)hipFree(pl->d_pad);
        (void)hipFree(pl->d_wpack);
        (void)hipFree(pl->d_tie_n);
        (void)hipFree(pl->d_cov_col);
        (void)hipFree(pl->d_cov_j0);
        (void)hipFree(pl->d_cov_cnt);
        (void)hipFree(pl->d_gv_pk);
        (void)hipFree(pl->d_gv_mult);
        (void)hipFree(pl->d_gv_col16);
        (void)hipFree(pl->d_row_list);
        (void)hipFree(pl->d_row_count);
        (void)hipFree(pl->d_cell_part);
        (void)hipFree(pl->d_chunk_part);
        (void)hipFree(pl->d_hb_stats);
        (void)hipFree(pl->d_win_scratch);
        (void)hipFree(pl->d_dst16);
        (void)hipFree(pl->d_x16_wdesc);
        (void)hipFree(pl->d_blk_g0);
        (void)hipFree(pl->d_se_w0);
        (void)hipFree(pl->d_se_w1);
        (void)hipFree(pl->d_zrow);
        if (pl->done_ev) (void)hipEventDestroy(pl->done_ev);
    }
    delete pl;
}

int icv_plan_get_info(icv_plan_t pl, icv_plan_info* info) {
    if (!pl || !info) return fail(ICV_ERR_INVALID, "null argument");
    const icv::Plan& p = pl->p;
    info->n_cols_all = p.n_cols_all;
    info->n_genes_used = p.n_used;
    info->n_chr = p.n_chr;
    info->window = p.window;
    info->step = p.step;
    info->n_windows = p.W;
    info->block = p.B;
    info->n_blocks = p.NB;
    info->padded_len = p.Gp;
    info->lds_bytes_f32 = p.lay32.total;
    info->lds_bytes_f64 = p.lay64.total;
    int per_cu = p.lay32.fits ? icv::kLdsLimit / p.lay32.total : 0;
    info->workgroups_per_cu_f32 = per_cu > 4 ? 4 : per_cu;
    return ICV_OK;
}

int icv_plan_last_kernel(icv_plan_t pl, int32_t* kind) {
    if (!pl || !kind) return fail(ICV_ERR_INVALID, "null argument");
    *kind = pl->last_kernel;
    return ICV_OK;
}

int icv_plan_chr_pos(icv_plan_t pl, int32_t* h_chr_pos) {
    if (!pl || !h_chr_pos) return fail(ICV_ERR_INVALID, "null argument");
    std::memcpy(h_chr_pos, pl->p.chr_pos.data(), pl->p.chr_pos.size() * sizeof(int32_t));
    return ICV_OK;
}

int icv_plan_window_table(icv_plan_t pl, int32_t* h_start, int32_t* h_len) {
    if (!pl || !h_start || !h_len) return fail(ICV_ERR_INVALID, "null argument");
    std::memcpy(h_start, pl->p.w_start_sorted.data(), pl->p.W * sizeof(int32_t));
    std::memcpy(h_len, pl->p.w_len_sorted.data(), pl->p.W * sizeof(int32_t));
    return ICV_OK;
}

int icv_plan_se_tables(icv_plan_t pl, int32_t* h_applies, int32_t* h_col_block, int32_t* h_col_offset,
                       int32_t* h_block_gene0, uint32_t* h_w0, uint32_t* h_w1) {
    if (!pl || !h_applies) return fail(ICV_ERR_INVALID, "null argument");
    const icv::Plan& p = pl->p;
    *h_applies = p.se_ok ? 1 : 0;
    if (!p.se_ok) return ICV_OK;
    if (h_col_block && h_col_offset)
        for (int g = 0; g < p.n_cols_all; ++g) {
            const int pos = p.dst[g];
            h_col_block[g] = pos < 0 ? -1 : pos / p.B;
            h_col_offset[g] = pos < 0 ? 0 : pos % p.B;
        }
    if (h_block_gene0) std::memcpy(h_block_gene0, p.blk_g0.data(), (size_t)p.NB * sizeof(int32_t));
    if (h_w0) std::memcpy(h_w0, p.se_w0.data(), (size_t)p.W * sizeof(uint32_t));
    if (h_w1) std::memcpy(h_w1, p.se_w1.data(), (size_t)p.W * sizeof(uint32_t));
    return ICV_OK;
}

int icv_plan_gene_runs(icv_plan_t pl, int32_t* h_n_runs, int32_t* h_run_first, int32_t* h_run_count,
                       int32_t* h_run_genes, int32_t* h_col_run) {
    if (!pl || !h_n_runs) return fail(ICV_ERR_INVALID, "null argument");
    const icv::Plan& p = pl->p;
    const size_t R = p.gv_run_j0.size();
    *h_n_runs = (int32_t)R;
    if (h_run_first) std::memcpy(h_run_first, p.gv_run_j0.data(), R * sizeof(int32_t));
    if (h_run_count) std::memcpy(h_run_count, p.gv_run_cnt.data(), R * sizeof(int32_t));
    if (h_run_genes) std::memcpy(h_run_genes, p.gv_run_mult.data(), R * sizeof(int32_t));
    if (h_col_run) std::memcpy(h_col_run, p.gv_col_run.data(), (size_t)p.n_cols_all * sizeof(int32_t));
    return ICV_OK;
}

int icv_colsum(const icv_matrix* m, const int32_t* row_group, int32_t n_groups, double* sums, void* stream) {
    if (!m || !sums || n_groups < 1) return fail(ICV_ERR_INVALID, "bad colsum arguments");
    if (m->n_rows == 0) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nc = m->n_cols;
    if (m->format == ICV_CSR) {
        // one 1024-thread workgroup per (slab, column tile), the whole LDS of a CU each: two slabs per CU
        int n_cu = 256;
        {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                n_cu = prop.multiProcessorCount;
        }
        int rows_per_slab = (int)((m->n_rows + (int64_t)n_cu * 2 - 1) / ((int64_t)n_cu * 2));
        if (rows_per_slab < 64) rows_per_slab = 64;
        const int64_t n_slabs = (m->n_rows + rows_per_slab - 1) / rows_per_slab;
        const int n_tiles = (nc + icv::kCsrTileCols - 1) / icv::kCsrTileCols;
        const int lds = (nc < icv::kCsrTileCols ? nc : icv::kCsrTileCols) * (int)sizeof(double);
        AsyncBuf partial_b;
        HIP_TRY(partial_b.alloc((size_t)n_slabs * nc * sizeof(double), st));
        double* partial = partial_b.as<double>();
        dim3 grid((unsigned)n_tiles, (unsigned)n_slabs), block(icv::kCsThreads);
        typedef void (*kf_t)(const float*, const int64_t*, const int32_t*, int64_t, int, const int32_t*, int, int, double*);
        typedef void (*kd_t)(const double*, const int64_t*, const int32_t*, int64_t, int, const int32_t*, int, int, double*);
        const kf_t kf = row_group ? (kf_t)icv::k_colsum_csr<float, true> : (kf_t)icv::k_colsum_csr<float, false>;
        const kd_t kd = row_group ? (kd_t)icv::k_colsum_csr<double, true> : (kd_t)icv::k_colsum_csr<double, false>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int g = 0; g < n_groups; ++g) {
            if (m->dtype == ICV_F32)
                hipLaunchKernelGGL(kf, grid, block, lds, st, (const float*)m->values, m->indptr, m->indices, m->n_rows,
                                   nc, row_group, g, rows_per_slab, partial);
            else
                hipLaunchKernelGGL(kd, grid, block, lds, st, (const double*)m->values, m->indptr, m->indices, m->n_rows,
                                   nc, row_group, g, rows_per_slab, partial);
            hipLaunchKernelGGL(icv::k_colsum_finish, dim3((nc + 63) / 64), dim3(1024), 0, st, partial, (int)n_slabs, nc,
                               sums + (int64_t)g * nc);
        }
        HIP_TRY(hipGetLastError());
        return ICV_OK;
    }
    // slab height from the row count alone (the sums must not depend on the device): 1024 rows stream fastest
    // (tools/microbench_colsum.hip), shorter slabs keep every CU busy on small matrices
    const int rows_per_slab = m->n_rows >= 65536 ? 1024 : 256;
    const int64_t n_slabs = (m->n_rows + rows_per_slab - 1) / rows_per_slab;
    AsyncBuf partial_b;
    HIP_TRY(partial_b.alloc((size_t)n_slabs * nc * sizeof(double), st));
    double* partial = partial_b.as<double>();
    dim3 grid((nc + 255) / 256, (unsigned)n_slabs), block(256);
    for (int g = 0; g < n_groups; ++g) {
        if (m->dtype == ICV_F32)
            hipLaunchKernelGGL(icv::k_colsum_dense<float>, grid, block, 0, st, (const float*)m->values, m->n_rows,
                               m->ld, nc, row_group, g, rows_per_slab, partial);
        else
            hipLaunchKernelGGL(icv::k_colsum_dense<double>, grid, block, 0, st, (const double*)m->values,
                               m->n_rows, m->ld, nc, row_group, g, rows_per_slab, partial);
        hipLaunchKernelGGL(icv::k_colsum_finish, dim3((nc + 63) / 64), dim3(1024), 0, st, partial, (int)n_slabs, nc,
                           sums + (int64_t)g * nc);
    }
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

// ---- the reference-order float32 chain by blocks (csrc/icv_kernel_blocks.hpp): row shards that do not take turns -----
namespace {
struct BlocksLayout {  // one workspace, laid out here (the caller allocates icv_colchain_blocks_workspace() bytes)
    int64_t n_slabs, n_blocks;
    size_t partial_off, start_off, slab_rec_off, rec_off, count_off, stash_off, bytes;
    unsigned stash_cap;
    BlocksLayout(int64_t n_rows, int32_t n_cols) {
        n_slabs = (n_rows + icv::kBkSlab - 1) / icv::kBkSlab;
        if (n_slabs < 1) n_slabs = 1;
        n_blocks = n_slabs * icv::kBkPerSlab;
        const size_t nc = (size_t)(n_cols > 0 ? n_cols : 1);
        auto up = [](size_t v) { return (v + 255) / 256 * 256; };
        partial_off = 0;
        start_off = up(partial_off + (size_t)n_slabs * nc * 8);
        slab_rec_off = up(start_off + (size_t)n_slabs * nc * 8);
        rec_off = up(slab_rec_off + (size_t)n_slabs * nc * 4);
        count_off = up(rec_off + (size_t)n_blocks * nc * 4);
        stash_off = up(count_off + 256);
        // stash: the blocks that cannot be summarised -- 0.1-1 % of the (block, column) pairs of a shard that starts
        // inside the chains, ~1 % + the first few thousand rows of a shard that starts them; 3 % here, the rest is
        // replayed from the matrix
        const size_t want = (size_t)n_blocks * nc / 32 + 4096;
        stash_cap = (unsigned)(want < 0x7ffffff0u ? want : 0x7ffffff0u);
        bytes = up(stash_off + (size_t)stash_cap * icv::kBkRows * 4);
    }
};
int blocks_check(const icv_matrix* m, const void* ws) {
    if (!m || !ws) return fail(ICV_ERR_INVALID, "null matrix or workspace");
    if (m->format != ICV_DENSE || m->dtype != ICV_F32)
        return fail(ICV_ERR_UNSUPPORTED, "the block form of the chain takes dense float32 matrices (others: icv_colchain)");
    if (m->n_rows < 0 || m->n_cols < 1 || m->ld < m->n_cols) return fail(ICV_ERR_INVALID, "bad matrix");
    return ICV_OK;
}
}  // namespace

int icv_colchain_blocks_workspace(int64_t n_rows, int32_t n_cols, int64_t* bytes) {
    if (!bytes || n_rows < 0 || n_cols < 1) return fail(ICV_ERR_INVALID, "bad colchain_blocks_workspace arguments");
    *bytes = (int64_t)BlocksLayout(n_rows, n_cols).bytes;
    return ICV_OK;
}

int icv_colchain_blocks_sums(const icv_matrix* m, void* workspace, double* total, void* stream) {
    int rc = blocks_check(m, workspace);
    if (rc) return rc;
    if (!total) return fail(ICV_ERR_INVALID, "total is required");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const BlocksLayout L(m->n_rows, m->n_cols);
    double* partial = reinterpret_cast<double*>(static_cast<char*>(workspace) + L.partial_off);
    HIP_TRY(hipMemsetAsync(total, 0, (size_t)m->n_cols * sizeof(double), st));
    if (m->n_rows == 0) return ICV_OK;
    dim3 grid((m->n_cols + 255) / 256, (unsigned)L.n_slabs);
    hipLaunchKernelGGL(icv::k_colsum_dense<float>, grid, dim3(256), 0, st, (const float*)m->values, m->n_rows, m->ld,
                       m->n_cols, (const int32_t*)nullptr, 0, icv::kBkSlab, partial);
    hipLaunchKernelGGL(icv::k_colsum_finish, dim3((m->n_cols + 63) / 64), dim3(1024), 0, st, partial, (int)L.n_slabs,
                       m->n_cols, total);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_colchain_blocks_records(const icv_matrix* m, void* workspace, const double* est_start, void* stream) {
    int rc = blocks_check(m, workspace);
    if (rc) return rc;
    if (m->n_rows == 0) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const BlocksLayout L(m->n_rows, m->n_cols);
    char* w = static_cast<char*>(workspace);
    const double* partial = reinterpret_cast<const double*>(w + L.partial_off);
    double* slab_start = reinterpret_cast<double*>(w + L.start_off);
    HIP_TRY(hipMemsetAsync(w + L.count_off, 0, 256, st));
    hipLaunchKernelGGL(icv::k_blocks_prefix, dim3((m->n_cols + 255) / 256), dim3(256), 0, st, partial, (int)L.n_slabs,
                       m->n_cols, est_start, slab_start);
    dim3 grid((m->n_cols + 255) / 256, (unsigned)L.n_slabs);
    hipLaunchKernelGGL(icv::k_chain_records, grid, dim3(256), 0, st, (const float*)m->values, m->n_rows, m->ld, m->n_cols,
                       slab_start, reinterpret_cast<uint32_t*>(w + L.rec_off), reinterpret_cast<uint32_t*>(w + L.slab_rec_off),
                       reinterpret_cast<float*>(w + L.stash_off), reinterpret_cast<unsigned*>(w + L.count_off), L.stash_cap);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_colchain_blocks_scan(const icv_matrix* m, void* workspace, float* acc, int32_t col0, int32_t col1,
                             uint64_t* d_replayed, void* stream) {
    int rc = blocks_check(m, workspace);
    if (rc) return rc;
    if (!acc || col0 < 0 || col1 > m->n_cols || col0 > col1) return fail(ICV_ERR_INVALID, "bad colchain_blocks_scan arguments");
    if (m->n_rows == 0 || col1 == col0) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const BlocksLayout L(m->n_rows, m->n_cols);
    char* w = static_cast<char*>(workspace);
    // (the column range: ranks pipeline the scan over column groups, as the chained form does)
    const int nc = col1 - col0;
    // one wavefront per column, four per workgroup, up to eight workgroups per CU at a time
    int64_t grid = ((int64_t)nc + 3) / 4;
    const int64_t cap = (int64_t)current_cu_count() * 8;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(icv::k_chain_scan, dim3((unsigned)grid), dim3(256), 0, st,
                       (const float*)m->values + col0, m->n_rows, m->ld, nc, (int64_t)m->n_cols,
                       reinterpret_cast<const uint32_t*>(w + L.rec_off) + col0,
                       reinterpret_cast<const uint32_t*>(w + L.slab_rec_off) + col0,
                       reinterpret_cast<const float*>(w + L.stash_off), acc + col0,
                       reinterpret_cast<unsigned long long*>(d_replayed));
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_colchain(const icv_matrix* m, const int32_t* rows, int64_t n_sel, double scale, void* acc, void* stream) {
    if (!m || !acc) return fail(ICV_ERR_INVALID, "bad colchain arguments");
    if (m->dtype != ICV_F32 && m->dtype != ICV_F64) return fail(ICV_ERR_INVALID, "dtype must be ICV_F32 or ICV_F64");
    if (!rows) n_sel = m->n_rows;
    if (n_sel < 0 || n_sel > m->n_rows) return fail(ICV_ERR_INVALID, "icv_colchain: n_sel out of range");
    if (n_sel == 0 || m->n_cols == 0) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (m->format == ICV_CSR)
        return m->dtype == ICV_F32 ? colchain_csr<float>(m, rows, n_sel, scale, (float*)acc, st)
                                   : colchain_csr<double>(m, rows, n_sel, scale, (double*)acc, st);
    if (m->format != ICV_DENSE) return fail(ICV_ERR_INVALID, "format must be dense or csr");
    return m->dtype == ICV_F32 ? colchain_dense<float>(m, rows, n_sel, (float*)acc, st)
                               : colchain_dense<double>(m, rows, n_sel, (double*)acc, st);
}

int icv_colchain_mean(const void* acc, int32_t dtype, int32_t n_cols, int64_t count, void* mean, void* stream) {
    if (!acc || !mean || count < 1 || n_cols < 0) return fail(ICV_ERR_INVALID, "bad colchain_mean arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_cols == 0) return ICV_OK;
    if (dtype == ICV_F32)
        hipLaunchKernelGGL(k_chain_mean<float>, dim3((n_cols + 255) / 256), dim3(256), 0, st, (const float*)acc, n_cols,
                           (float)count, (float*)mean);
    else if (dtype == ICV_F64)
        hipLaunchKernelGGL(k_chain_mean<double>, dim3((n_cols + 255) / 256), dim3(256), 0, st, (const double*)acc,
                           n_cols, (double)count, (double*)mean);
    else
        return fail(ICV_ERR_INVALID, "dtype must be ICV_F32 or ICV_F64");
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_colmean_csc(const void* values, int32_t dtype, const int64_t* colptr, const int32_t* row_idx, int32_t n_cols,
                    const int32_t* row_group, int32_t group, double scale, void* mean, void* stream) {
    if (!values || !colptr || !row_idx || !mean || n_cols < 0) return fail(ICV_ERR_INVALID, "bad colmean_csc arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_cols == 0) return ICV_OK;
    if (dtype == ICV_F32)
        hipLaunchKernelGGL(icv::k_colpair_csc<float>, dim3((n_cols + 63) / 64), dim3(64), 0, st, (const float*)values,
                           colptr, row_idx, n_cols, row_group, group, (float)scale, (float*)mean);
    else if (dtype == ICV_F64)
        hipLaunchKernelGGL(icv::k_colpair_csc<double>, dim3((n_cols + 63) / 64), dim3(64), 0, st, (const double*)values,
                           colptr, row_idx, n_cols, row_group, group, scale, (double*)mean);
    else
        return fail(ICV_ERR_INVALID, "dtype must be ICV_F32 or ICV_F64");
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_colsum_pairwise(const void* xt, int32_t dtype, int64_t n_rows, int32_t n_cols, int64_t ld, void* sums,
                        void* stream) {
    if (!xt || !sums || n_rows < 0 || n_cols < 0 || ld < n_rows) return fail(ICV_ERR_INVALID, "bad colsum_pairwise arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_cols == 0) return ICV_OK;
    if (dtype == ICV_F32)
        hipLaunchKernelGGL(icv::k_colpair_dense<float>, dim3((n_cols + 63) / 64), dim3(64), 0, st, (const float*)xt,
                           n_rows, n_cols, ld, (float*)sums);
    else if (dtype == ICV_F64)
        hipLaunchKernelGGL(icv::k_colpair_dense<double>, dim3((n_cols + 63) / 64), dim3(64), 0, st, (const double*)xt,
                           n_rows, n_cols, ld, (double*)sums);
    else
        return fail(ICV_ERR_INVALID, "dtype must be ICV_F32 or ICV_F64");
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_infercnv_smooth(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi,
                        double lfc_clip, int32_t flags, float* out, int64_t ldo, double* cell_median,
                        double* cell_stats, void* stream) {
    int rc = check_matrix(pl, m);
    if (rc) return rc;
    PLAN_BUSY_GUARD(pl);
    if (!cell_median || !cell_stats) return fail(ICV_ERR_INVALID, "cell_median and cell_stats are required");
    if ((rc = ensure_device(pl))) return rc;
    PLAN_ENTER(stream);
    icv::KParams K;
    const icv::Layout* lay;
    if ((rc = fill_params(pl, m, ref_lo, ref_hi, lfc_clip, flags, out, ldo, cell_median, cell_stats, K, lay)))
        return rc;
    return launch_smooth(pl, m, K, *lay, static_cast<hipStream_t>(stream));
}

int icv_chunk_thresholds(const double* cell_stats, int64_t n_rows, int64_t chunksize, int64_t row_phase,
                         int32_t n_windows, double dynamic_threshold, double* thr, void* stream) {
    if (!cell_stats || !thr || chunksize < 1 || row_phase < 0 || row_phase >= chunksize)
        return fail(ICV_ERR_INVALID, "bad chunk threshold arguments");
    if (n_rows < 1) return ICV_OK;
    const int64_t n_chunks = (n_rows + row_phase + chunksize - 1) / chunksize;
    hipLaunchKernelGGL(icv::k_chunk_thr, dim3((unsigned)n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       cell_stats, n_rows, chunksize, row_phase, n_windows, dynamic_threshold, thr);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_apply_threshold(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi,
                        double lfc_clip, int32_t flags, float* out, int64_t ldo, const double* cell_median,
                        const double* thr, int64_t chunksize, int64_t row_phase, void* stream) {
    int rc = check_matrix(pl, m);
    if (rc) return rc;
    PLAN_BUSY_GUARD(pl);
    if (!cell_median || !thr || chunksize < 1) return fail(ICV_ERR_INVALID, "bad apply_threshold arguments");
    if ((rc = ensure_device(pl))) return rc;
    PLAN_ENTER(stream);
    icv::KParams K;
    const icv::Layout* lay;
    if ((rc = fill_params(pl, m, ref_lo, ref_hi, lfc_clip, flags, out, ldo, const_cast<double*>(cell_median),
                          nullptr, K, lay)))
        return rc;
    return launch_apply(m, K, thr, chunksize, row_phase, static_cast<hipStream_t>(stream));
}

}  // extern "C"

namespace {
// steps 1-5 of the chunk kernel (icv_infercnv_run); win_out != nullptr: the float64 windows before centring as well
// (icv_infercnv_run_windows).  The caller holds the plan (PLAN_BUSY_GUARD + PLAN_ENTER).
int run_steps(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi, double lfc_clip,
              double dynamic_threshold, int64_t chunksize, int64_t row_phase, int32_t flags, float* out, int64_t ldo,
              double* cell_median, double* cell_stats, double* thr, icv_profile* prof, double* win_out, int64_t ldw,
              void* stream) {
    int rc;
    const bool do_thr = !std::isnan(dynamic_threshold);
    hipStream_t st = static_cast<hipStream_t>(stream);
    icv::KParams K;
    const icv::Layout* lay;
    if ((rc = fill_params(pl, m, ref_lo, ref_hi, lfc_clip, flags, out, ldo, cell_median, cell_stats, K, lay)))
        return rc;
    K.win_out = win_out;
    K.win_ld = ldw;
    // cell_stats == NULL: the caller only wants the thresholds.  The per-row moments then live in a plan-owned
    // buffer, and where k_smooth_x16 runs they are not formed per cell at all: the kernel accumulates them per
    // chunk (one wavefront reduction per chunk instead of per cell).
    bool chunk_mode = false;
    int64_t n_chunks = 1, cs_eff = m->n_rows > 0 ? m->n_rows : 1, ph_eff = 0;
    if (do_thr) {
        cs_eff = chunksize;
        ph_eff = row_phase;
        n_chunks = (m->n_rows + row_phase + chunksize - 1) / chunksize;
        if (n_chunks < 1) n_chunks = 1;
    }
    if (!cell_stats && m->n_rows > 0) {
        if (pl->hb_stats_cap < m->n_rows) {
            (void)hipFree(pl->d_hb_stats);
            pl->d_hb_stats = nullptr;
            pl->hb_stats_cap = 0;
            HIP_TRY(hipMalloc((void**)&pl->d_hb_stats, (size_t)m->n_rows * 2 * sizeof(double)));
            pl->hb_stats_cap = m->n_rows;
        }
        K.cell_stats = pl->d_hb_stats;
        // partial-moment slots per chunk: k_smooth_x16: n_cu workgroups x 16 wavefronts; k_smooth_se: its own grid
        // (se_grid: whatever its LDS map allows per CU) x 8 wavefronts -- sized for the larger of the two
        const int64_t se_slots = ((int64_t)pl->n_cu * (icv::kLdsLimit / icv::kSeLds)) * icv::NWAVE;
        const int64_t x16_slots = (int64_t)pl->n_cu * icv::XWAVE;
        const int64_t n_part = se_slots > x16_slots ? se_slots : x16_slots;
        if ((x16_applies(pl, m, K, *lay) || stored_entries_kernel(pl, m, K, *lay)) &&
            n_chunks * n_part <= (int64_t)(64 << 20) / 16) {
            chunk_mode = true;
            if (pl->chunk_part_cap < n_chunks * n_part) {
                (void)hipFree(pl->d_chunk_part);
                pl->d_chunk_part = nullptr;
                pl->chunk_part_cap = 0;
                HIP_TRY(hipMalloc((void**)&pl->d_chunk_part, (size_t)(n_chunks * n_part) * 2 * sizeof(double)));
                pl->chunk_part_cap = n_chunks * n_part;
            }
            HIP_TRY(hipMemsetAsync(pl->d_chunk_part, 0, (size_t)(n_chunks * n_part) * 2 * sizeof(double), st));
            HIP_TRY(hipMemsetAsync(pl->d_hb_stats, 0, (size_t)m->n_rows * 2 * sizeof(double), st));
            K.chunk_part = pl->d_chunk_part;
            K.chunksize = cs_eff;
            K.row_phase = ph_eff;
        }
    }
    EventSet evs;
    hipEvent_t* ev = evs.ev;
    const bool deferred = !prof && pl->prof_deferred;
    const bool timed = prof || deferred;
    if (timed) {
        for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreate(&ev[i]));
        HIP_TRY(hipEventRecord(ev[0], st));
    }
    bool ev1_done = false;
    if ((rc = launch_smooth(pl, m, K, *lay, st, timed ? ev[1] : nullptr, &ev1_done))) return rc;
    if (timed && !ev1_done) HIP_TRY(hipEventRecord(ev[1], st));
    if (do_thr && chunk_mode) {
        // partial slots of the launch: min(CUs, rows) workgroups x 16 wavefronts (k_smooth_x16), min(2 CUs, rows) x 8
        // (k_smooth_se); slots of absent workgroups are zero
        int64_t n_slots;
        if (m->format == ICV_CSR) {
            n_slots = se_grid(pl, m->n_rows) * icv::NWAVE;  // the grid launch_smooth_se used
        } else {
            int64_t gx = pl->n_cu;
            if (gx > m->n_rows) gx = m->n_rows;
            n_slots = gx * icv::XWAVE;
        }
        hipLaunchKernelGGL(icv::k_chunk_thr_part, dim3((unsigned)n_chunks), dim3(256), 0, st, pl->d_chunk_part,
                           (int)n_slots, pl->d_hb_stats, m->n_rows, chunksize, row_phase, pl->p.W,
                           dynamic_threshold, thr);
        HIP_TRY(hipGetLastError());
    } else if (do_thr && m->n_rows > 0) {
        rc = icv_chunk_thresholds(K.cell_stats, m->n_rows, chunksize, row_phase, pl->p.W, dynamic_threshold, thr,
                                  stream);
        if (rc) return rc;
    }
    if (timed) HIP_TRY(hipEventRecord(ev[2], st));
    if (do_thr && !(flags & ICV_FLAG_NO_APPLY) && (rc = launch_apply(m, K, thr, chunksize, row_phase, st))) return rc;
    if (timed) HIP_TRY(hipEventRecord(ev[3], st));
    if (deferred) {  // no synchronisation here: the times are read by icv_profile_collect
        for (int i = 0; i < 4; ++i) pl->prof_events.push_back(ev[i]);
        evs.keep = true;
    } else if (prof) {
        HIP_TRY(hipEventSynchronize(ev[3]));
        HIP_TRY(hipEventElapsedTime(&prof->smooth_ms, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&prof->thresholds_ms, ev[1], ev[2]));
        HIP_TRY(hipEventElapsedTime(&prof->apply_ms, ev[2], ev[3]));
        HIP_TRY(hipEventElapsedTime(&prof->total_ms, ev[0], ev[3]));
    }
    return ICV_OK;
}

// gene values of n rows from their float64 windows: the fused kernel where a cell's windows and run values fit LDS,
// the three round-1 kernels on a cells x n_cov temporary otherwise
int gene_from_windows(icv_plan_t pl, const double* win, int64_t ldw, int64_t n, const double* thr, int64_t chunksize,
                      int64_t row_phase, double* gene_out, int64_t ldg, hipStream_t st) {
    const icv::Plan& p = pl->p;
    if (n < 1) return ICV_OK;
    const int W = p.W, n_cov = (int)p.cov_col.size(), R = (int)p.gv_run_j0.size();
    // the run table in LDS too where two workgroups per CU still fit (87 VGPRs allow no more than two anyway)
    const bool pk_lds = icv::gv_lds_bytes(W, R, pl->gv_mult_bytes, true) * 2 <= (size_t)icv::kLdsLimit;
    const size_t lds = icv::gv_lds_bytes(W, R, pl->gv_mult_bytes, pk_lds);
    if (lds <= (size_t)icv::kLdsLimit && pl->gv_fused_ok && !knobs().no_gene_fused) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(icv::k_gene_fused),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = (int)((size_t)icv::kLdsLimit / lds);
        if (per_cu > 2048 / icv::kGvThreads) per_cu = 2048 / icv::kGvThreads;  // 32 wavefronts per CU
        int64_t grid = (int64_t)pl->n_cu * per_cu;
        if (grid > n) grid = n;
        hipLaunchKernelGGL(icv::k_gene_fused, dim3((unsigned)grid), dim3(icv::kGvThreads), lds, st, win, ldw, n, W,
                           pl->d_gv_pk, pl->d_gv_mult, R, n_cov, pl->d_gv_col16, p.n_cols_all, thr,
                           chunksize > 0 ? chunksize : 1, row_phase, gene_out, ldg, pl->gv_mult_bytes, (int)pk_lds);
        HIP_TRY(hipGetLastError());
        return ICV_OK;
    }
    AsyncBuf b_gv, b_med;
    HIP_TRY(b_gv.alloc((size_t)n * (n_cov > 0 ? n_cov : 1) * sizeof(double), st));
    HIP_TRY(b_med.alloc((size_t)n * sizeof(double), st));
    double *gv = b_gv.as<double>(), *med = b_med.as<double>();
    {
        const int64_t total = n * ldg;
        hipLaunchKernelGGL(icv::k_fill_nan, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gene_out, total);
    }
    if (n_cov > 0) {
        for (int64_t r0 = 0; r0 < n; r0 += 32768) {
            const int64_t nr = (n - r0) < 32768 ? (n - r0) : 32768;
            dim3 grid((n_cov + 255) / 256, (unsigned)nr);
            hipLaunchKernelGGL(icv::k_gene_means, grid, dim3(256), 0, st, win + r0 * ldw, nr, (int)ldw, pl->d_cov_j0,
                               pl->d_cov_cnt, n_cov, gv + r0 * n_cov);
        }
        hipLaunchKernelGGL(icv::k_row_median, dim3((unsigned)n), dim3(256), 0, st, gv, n, n_cov, med);
        for (int64_t r0 = 0; r0 < n; r0 += 32768) {
            const int64_t nr = (n - r0) < 32768 ? (n - r0) : 32768;
            dim3 grid((n_cov + 255) / 256, (unsigned)nr);
            // chunk of row r0 + i: (r0 + i + row_phase) / chunksize relative to thr[0]
            hipLaunchKernelGGL(icv::k_gene_finish, grid, dim3(256), 0, st, gv + r0 * n_cov, med + r0, thr,
                               chunksize > 0 ? chunksize : 1, row_phase + r0, pl->d_cov_col, n_cov, gene_out + r0 * ldg,
                               ldg);
        }
    }
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}
}  // namespace

extern "C" {

int icv_infercnv_run(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi, double lfc_clip,
                     double dynamic_threshold, int64_t chunksize, int64_t row_phase, int32_t flags, float* out,
                     int64_t ldo, double* cell_median, double* cell_stats, double* thr, icv_profile* prof,
                     void* stream) {
    return icv_infercnv_run_windows(pl, m, ref_lo, ref_hi, lfc_clip, dynamic_threshold, chunksize, row_phase, flags, out,
                                    ldo, cell_median, cell_stats, thr, prof, nullptr, 0, stream);
}

int icv_infercnv_run_windows(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi,
                             double lfc_clip, double dynamic_threshold, int64_t chunksize, int64_t row_phase,
                             int32_t flags, float* out, int64_t ldo, double* cell_median, double* cell_stats,
                             double* thr, icv_profile* prof, double* win_out, int64_t ldw, void* stream) {
    int rc = check_matrix(pl, m);
    if (rc) return rc;
    PLAN_BUSY_GUARD(pl);
    const bool do_thr = !std::isnan(dynamic_threshold);
    if (m->n_rows == 0) return ICV_OK;  // an empty shard: no cell, no chunk, nothing written
    if (!cell_median) return fail(ICV_ERR_INVALID, "cell_median is required");
    if (do_thr && (!thr || chunksize < 1 || row_phase < 0 || row_phase >= chunksize))
        return fail(ICV_ERR_INVALID, "thr buffer / chunksize / row_phase invalid");
    if (win_out && ldw < pl->p.W) return fail(ICV_ERR_INVALID, "ldw < n_windows");
    if ((rc = ensure_device(pl))) return rc;
    PLAN_ENTER(stream);
    return run_steps(pl, m, ref_lo, ref_hi, lfc_clip, dynamic_threshold, chunksize, row_phase, flags, out, ldo,
                     cell_median, cell_stats, thr, prof, win_out, ldw, stream);
}

int icv_gene_values_from_windows(icv_plan_t pl, const double* win, int64_t ldw, int64_t n_rows, const double* thr,
                                 int64_t chunksize, int64_t row_phase, double* gene_out, int64_t ldg, void* stream) {
    if (!pl) return fail(ICV_ERR_INVALID, "null plan");
    PLAN_BUSY_GUARD(pl);
    if (n_rows < 0 || (n_rows > 0 && (!win || !gene_out)) || ldw < pl->p.W || ldg < pl->p.n_cols_all)
        return fail(ICV_ERR_INVALID, "bad gene_values_from_windows arguments (null buffer, ldw < n_windows or ldg < n_cols)");
    if (thr && (chunksize < 1 || row_phase < 0 || row_phase >= chunksize))
        return fail(ICV_ERR_INVALID, "chunksize / row_phase invalid");
    int rc;
    if ((rc = ensure_device(pl))) return rc;
    PLAN_ENTER(stream);
    return gene_from_windows(pl, win, ldw, n_rows, thr, chunksize, row_phase, gene_out, ldg,
                             static_cast<hipStream_t>(stream));
}

int icv_profile_begin(icv_plan_t pl) {
    if (!pl) return fail(ICV_ERR_INVALID, "null plan");
    for (auto& e : pl->prof_events) (void)hipEventDestroy(e);
    pl->prof_events.clear();
    pl->prof_deferred = true;
    return ICV_OK;
}

int icv_profile_collect(icv_plan_t pl, icv_profile* out, int32_t max_records, int32_t* n_records) {
    if (!pl || !n_records || (max_records > 0 && !out)) return fail(ICV_ERR_INVALID, "bad profile_collect arguments");
    const int n = (int)(pl->prof_events.size() / 4);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        hipEvent_t* ev = pl->prof_events.data() + 4 * i;
        if (k < max_records) {
            HIP_TRY(hipEventSynchronize(ev[3]));
            HIP_TRY(hipEventElapsedTime(&out[k].smooth_ms, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&out[k].thresholds_ms, ev[1], ev[2]));
            HIP_TRY(hipEventElapsedTime(&out[k].apply_ms, ev[2], ev[3]));
            HIP_TRY(hipEventElapsedTime(&out[k].total_ms, ev[0], ev[3]));
            ++k;
        }
    }
    for (auto& e : pl->prof_events) (void)hipEventDestroy(e);
    pl->prof_events.clear();
    pl->prof_deferred = false;
    *n_records = k;
    return ICV_OK;
}

int icv_gene_values(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi, double lfc_clip,
                    int32_t flags, const double* thr, int64_t chunksize, int64_t row_phase, double* gene_out,
                    int64_t ldg, void* stream) {
    int rc = check_matrix(pl, m);
    if (rc) return rc;
    PLAN_BUSY_GUARD(pl);
    if (!gene_out || ldg < m->n_cols) return fail(ICV_ERR_INVALID, "gene_out is null or ldg < n_cols");
    if (thr && (chunksize < 1 || row_phase < 0 || row_phase >= chunksize))
        return fail(ICV_ERR_INVALID, "chunksize / row_phase invalid");
    if ((rc = ensure_device(pl))) return rc;
    PLAN_ENTER(stream);
    const int64_t n = m->n_rows;
    if (n < 1) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int W = pl->p.W;
    // ONE smoothing pass through the kernel the plain call takes (k_smooth_x16 / k_smooth_se write the float64 windows
    // beside x_res; other geometries: the generic kernel), then the fused gene kernel.  (A caller that also wants
    // X_cnv uses icv_infercnv_run_windows + icv_gene_values_from_windows: no second smoothing at all.)
    AsyncBuf b_out32, b_win, b_cmed;
    HIP_TRY(b_out32.alloc((size_t)n * W * sizeof(float), st));
    HIP_TRY(b_win.alloc((size_t)n * W * sizeof(double), st));
    HIP_TRY(b_cmed.alloc((size_t)n * sizeof(double), st));
    if ((rc = run_steps(pl, m, ref_lo, ref_hi, lfc_clip, std::nan(""), 1, 0, flags, b_out32.as<float>(), W,
                        b_cmed.as<double>(), nullptr, nullptr, nullptr, b_win.as<double>(), W, stream)))
        return rc;
    return gene_from_windows(pl, b_win.as<double>(), W, n, thr, chunksize, row_phase, gene_out, ldg, st);
}

int icv_threshold_mask(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi, double lfc_clip,
                       int32_t flags, const float* out, int64_t ldo, const double* cell_median, const double* thr,
                       int64_t chunksize, int64_t row_phase, uint64_t* mask, int64_t* row_nnz, void* stream) {
    int rc = check_matrix(pl, m);
    if (rc) return rc;
    PLAN_BUSY_GUARD(pl);
    if (!cell_median || !mask || !row_nnz || (thr && (chunksize < 1 || row_phase < 0 || row_phase >= chunksize)))
        return fail(ICV_ERR_INVALID, "bad threshold_mask arguments");
    if ((rc = ensure_device(pl))) return rc;
    PLAN_ENTER(stream);
    icv::KParams K;
    const icv::Layout* lay;
    if ((rc = fill_params(pl, m, ref_lo, ref_hi, lfc_clip, flags, const_cast<float*>(out), ldo,
                          const_cast<double*>(cell_median), nullptr, K, lay)))
        return rc;
    if (K.n_rows < 1) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int n_words = (pl->p.W + 63) / 64;
    const int64_t cs = thr ? chunksize : 1;
    auto* mk = reinterpret_cast<unsigned long long*>(mask);
    // streamed form (k_thr_mask_ring: rows through an LDS ring by LDS-DMA, one persistent workgroup per CU) where the
    // rows are 16-byte aligned and at least 1 KB; else one workgroup per row
    const icv::PackRing pr(pl->p.W);
    const bool ring = !knobs().no_mask_ring && pr.ok && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(mask) & 7) == 0;
    if (ring) {
        const int lds = pr.lds_bytes();
        const int64_t rounds = (K.n_rows + icv::kPmRows - 1) / icv::kPmRows;
        const unsigned gx = (unsigned)(rounds < pl->n_cu ? rounds : pl->n_cu);
        AsyncBuf ties;  // (row, wavefront part) pairs whose windows float32 could not decide
        if (thr) HIP_TRY(ties.alloc((size_t)K.n_rows * icv::kPmPerRow * sizeof(unsigned long long), st));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(icv::k_thr_mask_ring),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, icv::kPmLds));  // (per device: every call)
        hipLaunchKernelGGL(icv::k_thr_mask_ring, dim3(gx), dim3(icv::kPmThreads), lds, st, K, thr, cs, row_phase, mk, n_words,
                           row_nnz, pl->d_tie_n, ties.as<unsigned long long>());
        if (thr) {
            dim3 tg(128), tb(256);
            const auto* tl = ties.as<unsigned long long>();
            if (m->dtype == ICV_F32) {
                if (m->format == ICV_DENSE)
                    hipLaunchKernelGGL((icv::k_thr_mask_ties<float, false>), tg, tb, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz, pl->d_tie_n, tl);
                else
                    hipLaunchKernelGGL((icv::k_thr_mask_ties<float, true>), tg, tb, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz, pl->d_tie_n, tl);
            } else {
                if (m->format == ICV_DENSE)
                    hipLaunchKernelGGL((icv::k_thr_mask_ties<double, false>), tg, tb, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz, pl->d_tie_n, tl);
                else
                    hipLaunchKernelGGL((icv::k_thr_mask_ties<double, true>), tg, tb, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz, pl->d_tie_n, tl);
            }
        }
        HIP_TRY(hipGetLastError());
        return ICV_OK;
    }
    dim3 grid((unsigned)K.n_rows), block(256);
    if (m->dtype == ICV_F32) {
        if (m->format == ICV_DENSE)
            hipLaunchKernelGGL((icv::k_thr_mask<float, false>), grid, block, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz);
        else
            hipLaunchKernelGGL((icv::k_thr_mask<float, true>), grid, block, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz);
    } else {
        if (m->format == ICV_DENSE)
            hipLaunchKernelGGL((icv::k_thr_mask<double, false>), grid, block, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz);
        else
            hipLaunchKernelGGL((icv::k_thr_mask<double, true>), grid, block, 0, st, K, thr, cs, row_phase, mk, n_words, row_nnz);
    }
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_threshold_pack(icv_plan_t pl, const icv_matrix* m, const void* ref_lo, const void* ref_hi, double lfc_clip,
                       int32_t flags, const float* out, int64_t ldo, const double* cell_median, const double* thr,
                       int64_t chunksize, int64_t row_phase, int64_t* indptr, int32_t* indices, double* data,
                       int64_t capacity, void* stream) {
    int rc = check_matrix(pl, m);
    if (rc) return rc;
    PLAN_BUSY_GUARD(pl);
    if (!cell_median || !indptr || !indices || !data || capacity < 0 ||
        (thr && (chunksize < 1 || row_phase < 0 || row_phase >= chunksize)))
        return fail(ICV_ERR_INVALID, "bad threshold_pack arguments");
    if (pl->p.W > 320 * 64) return fail(ICV_ERR_UNSUPPORTED, "icv_threshold_pack: more than 20480 windows");
    // rows per workgroup (= per look-back ticket): as many as keep their mask words in the workgroup's LDS, at most 16
    const int n_words = (pl->p.W + 63) / 64;
    int rpw = icv::kPackWords / n_words;
    rpw = rpw < 1 ? 1 : (rpw > icv::kPackMaxRows ? icv::kPackMaxRows : rpw);
    if ((rc = ensure_device(pl))) return rc;
    PLAN_ENTER(stream);
    icv::KParams K;
    const icv::Layout* lay;
    if ((rc = fill_params(pl, m, ref_lo, ref_hi, lfc_clip, flags, const_cast<float*>(out), ldo,
                          const_cast<double*>(cell_median), nullptr, K, lay)))
        return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (K.n_rows < 1) {
        HIP_TRY(hipMemsetAsync(indptr, 0, sizeof(int64_t), st));
        return ICV_OK;
    }
    if (K.n_rows > 0xffffffffll) return fail(ICV_ERR_UNSUPPORTED, "icv_threshold_pack: more than 2^32 rows per call");
    // look-back workspace: one status word per ticket, two per group of 64 tickets, the ticket counter; zeroed in
    // stream order
    const int64_t n_tickets = (K.n_rows + rpw - 1) / rpw, n_groups = (n_tickets + 63) / 64;
    AsyncBuf ws;
    HIP_TRY(ws.alloc((size_t)(n_tickets + 2 * n_groups + 1) * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(ws.p, 0, (size_t)(n_tickets + 2 * n_groups + 1) * sizeof(unsigned long long), st));
    auto* status = ws.as<unsigned long long>();
    auto* gstat = status + n_tickets;
    auto* gacc = gstat + n_groups;
    auto* ticket = reinterpret_cast<unsigned int*>(gacc + n_groups);
    const int64_t cs = thr ? chunksize : 1;
    dim3 grid((unsigned)n_tickets), block(256);
#define ICV_PACK(TT, CC) \
    hipLaunchKernelGGL((icv::k_thr_pack<TT, CC>), grid, block, 0, st, K, thr, cs, row_phase, rpw, ticket, status, gstat, gacc, indptr, indices, data, capacity)
    if (m->dtype == ICV_F32) {
        if (m->format == ICV_DENSE) ICV_PACK(float, false);
        else ICV_PACK(float, true);
    } else {
        if (m->format == ICV_DENSE) ICV_PACK(double, false);
        else ICV_PACK(double, true);
    }
#undef ICV_PACK
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_pack_geometry(int32_t n_windows, icv_pack_info* info) {
    if (!info || n_windows < 1) return fail(ICV_ERR_INVALID, "bad pack_geometry arguments");
    const icv::PackRing pr(n_windows);
    const icv::FillRing fr(n_windows);
    info->rows_per_round = icv::kPmRows;
    info->mask_streamed = pr.ok ? 1 : 0;
    info->mask_ring_slots = pr.ok ? pr.n_slots : 0;
    info->mask_lds_bytes = pr.ok ? pr.lds_bytes() : 0;
    info->mask_loads_in_flight = pr.ok ? (pr.n_slots - 1) * pr.per_loader : 0;
    info->fill_streamed = fr.ok ? 1 : 0;
    info->fill_ring_slots = fr.ok ? fr.n_slots : 0;
    info->fill_lds_bytes = fr.ok ? fr.lds_bytes() : 0;
    info->fill_loads_in_flight = fr.ok ? (fr.n_slots - 1) * (fr.per_loader + fr.n_extra) : 0;
    info->fill_stage_entries = fr.ok ? fr.cap : 0;
    return ICV_OK;
}

int icv_row_offsets(const int64_t* row_nnz, int64_t n_rows, int64_t* indptr, void* stream) {
    if ((!row_nnz && n_rows > 0) || !indptr || n_rows < 0) return fail(ICV_ERR_INVALID, "bad row_offsets arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned nb = (unsigned)((n_rows + icv::kScanBlock - 1) / icv::kScanBlock);
    AsyncBuf sums;
    HIP_TRY(sums.alloc((size_t)(nb ? nb : 1) * sizeof(int64_t), st));
    if (nb > 1) hipLaunchKernelGGL(icv::k_row_block_sums, dim3(nb), dim3(1024), 0, st, row_nnz, n_rows, sums.as<int64_t>());
    hipLaunchKernelGGL(icv::k_row_offsets, dim3(nb ? nb : 1), dim3(1024), 0, st, row_nnz, n_rows, sums.as<int64_t>(), indptr);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_csr_fill_masked(const float* x, int64_t n_rows, int32_t n_cols, int64_t ld, const uint64_t* mask,
                        const int64_t* indptr, int32_t* indices, double* data, void* stream) {
    if (!x || !mask || !indptr || !indices || !data || n_cols < 0 || ld < n_cols)
        return fail(ICV_ERR_INVALID, "bad csr_fill_masked arguments");
    if (n_rows < 1) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // streamed form (k_csr_fill_ring: rows, mask rows and row offsets through an LDS ring, one persistent workgroup per
    // CU) where the rows are 16-byte aligned and at least 1 KB; else one wavefront per row
    const icv::FillRing fr(n_cols);
    if (!knobs().no_fill_ring && fr.ok && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(mask) & 7) == 0) {
        const int n_cu = current_cu_count();
        const int64_t rounds = (n_rows + icv::kPmRows - 1) / icv::kPmRows;
        const unsigned gx = (unsigned)(rounds < n_cu ? rounds : n_cu);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(icv::k_csr_fill_ring),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, icv::kPmLds));  // (per device: every call)
        hipLaunchKernelGGL(icv::k_csr_fill_ring, dim3(gx), dim3(icv::kPmThreads), fr.lds_bytes(), st, x, n_rows, n_cols, ld,
                           reinterpret_cast<const unsigned long long*>(mask), indptr, indices, data);
        HIP_TRY(hipGetLastError());
        return ICV_OK;
    }
    hipLaunchKernelGGL(icv::k_csr_fill_masked, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, x, n_rows, n_cols, ld,
                       reinterpret_cast<const unsigned long long*>(mask), (n_cols + 63) / 64, indptr, indices, data);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_csr_count(const float* x, int64_t n_rows, int32_t n_cols, int64_t ld, int64_t* row_nnz, void* stream) {
    if (!x || !row_nnz || n_cols < 0 || ld < n_cols) return fail(ICV_ERR_INVALID, "bad csr_count arguments");
    if (n_rows < 1) return ICV_OK;
    hipLaunchKernelGGL(icv::k_csr_count, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n_rows, n_cols, ld, row_nnz);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_csr_fill(const float* x, int64_t n_rows, int32_t n_cols, int64_t ld, const int64_t* indptr, int32_t* indices,
                 double* data, void* stream) {
    if (!x || !indptr || !indices || !data || n_cols < 0 || ld < n_cols)
        return fail(ICV_ERR_INVALID, "bad csr_fill arguments");
    if (n_rows < 1) return ICV_OK;
    hipLaunchKernelGGL(icv::k_csr_fill, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n_rows, n_cols, ld, indptr, indices, data);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

}  // extern "C"

// ---- Gram / distance tile jobs (icv_corr.hpp) -------------------------------------------------------
namespace {
struct GramWork {  // device side of one k_gram_mfma launch: the super-tile descriptors
    char* buf = nullptr;
    ~GramWork() { (void)hipFree(buf); }
};

template <bool DIST, bool SYM>
int launch_gram(hipStream_t st, GramWork& w, const float* z, int kz, const double* norm,
                const std::vector<icv::GramSuper>& supers, int64_t row_end, int64_t col_end, float* c_dir,
                int64_t ld_dir, float* c_mir, int64_t ld_mir) {
    if (supers.empty()) return ICV_OK;
    HIP_TRY(hipMalloc((void**)&w.buf, supers.size() * sizeof(icv::GramSuper)));
    HIP_TRY(hipMemcpyAsync(w.buf, supers.data(), supers.size() * sizeof(icv::GramSuper), hipMemcpyHostToDevice, st));
    icv::GramJob J;
    J.supers = reinterpret_cast<const icv::GramSuper*>(w.buf);
    J.n_supers = (int)supers.size();
    J.row_end = row_end;
    J.col_end = col_end;
    J.c_dir = c_dir;
    J.ld_dir = ld_dir;
    J.c_mir = c_mir;
    J.ld_mir = ld_mir;
    const unsigned grid = 512u * (unsigned)((supers.size() + 7) / 8);
    hipLaunchKernelGGL((icv::k_gram_mfma<DIST, SYM>), dim3(grid), dim3(256), 0, st, z, kz, norm, J);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

// super-tiles of the full symmetric n x n result (upper triangle, row-major)
std::vector<icv::GramSuper> gram_supers_sym(int64_t n, int64_t ld) {
    const int64_t ss = (int64_t)icv::GT * icv::GSUPER;
    const int64_t ns = (n + ss - 1) / ss;
    std::vector<icv::GramSuper> v;
    v.reserve((size_t)(ns * (ns + 1) / 2));
    for (int64_t sy = 0; sy < ns; ++sy)
        for (int64_t sx = sy; sx < ns; ++sx)
            v.push_back({(int)(sy * ss), (int)(sx * ss), sy * ss * ld + sx * ss, sx * ss * ld + sy * ss});
    return v;
}
}  // namespace

extern "C" {

int icv_corr_iqr(const float* x, int64_t n, int32_t k, int64_t ld, double* h_iqr, void* stream) {
    if (!x || !h_iqr || n < 2 || k < 1 || ld < k) return fail(ICV_ERR_INVALID, "bad corr_iqr arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int kz = icv::round_up(k, icv::GK);
    // stream-ordered temporaries from the device's pool (kept between calls up to AsyncBuf's threshold: a fresh
    // hipMalloc of the n x n matrix costs tens of milliseconds on a box whose VRAM has not been touched yet)
    AsyncBuf z_b, c_b, cnt_b;
    HIP_TRY(z_b.alloc((size_t)n * kz * sizeof(float), st));
    HIP_TRY(c_b.alloc((size_t)n * n * sizeof(float), st));
    HIP_TRY(cnt_b.alloc((4 * 2048 + 1) * sizeof(unsigned long long), st));
    float *z = z_b.as<float>(), *c = c_b.as<float>();
    unsigned long long* d_cnt = cnt_b.as<unsigned long long>();
    hipLaunchKernelGGL(icv::k_row_normalize, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x, n, k, ld, z, kz);
    GramWork gw;
    if (int rc = launch_gram<false, true>(st, gw, z, kz, nullptr, gram_supers_sym(n, n), n, n, c, n, c, n)) return rc;
    // the 25 % and 75 % percentiles interpolate between order statistics floor(pos), floor(pos) + 1
    const double m1 = (double)n * (double)n - 1.0;
    const double pos[2] = {0.25 * m1, 0.75 * m1};
    unsigned long long target[4];
    for (int q = 0; q < 2; ++q) {
        target[2 * q] = (unsigned long long)std::floor(pos[q]);
        target[2 * q + 1] = target[2 * q] + 1 < (unsigned long long)(m1 + 1.0) ? target[2 * q] + 1 : target[2 * q];
    }
    // the keys of ranks target[0..3] by radix selection on the ordered 32-bit keys (exact): 11 + 11 + 10 bits, one pass
    // over the matrix each (k_key_hist; targets that share their higher bits share a histogram).  Round 5 bisected with
    // k_count_le4: 33 passes, half of ithcna's time at 25 000 cells per group.
    unsigned lo[4] = {0, 0, 0, 0};
    const int64_t m = n * n;
    int64_t grid = (m + 255) / 256;
    if (grid > 4096) grid = 4096;
    bool any_nan = false;
    {
        std::vector<unsigned long long> h(4 * 2048 + 1);
        unsigned long long below[4] = {0, 0, 0, 0};  // elements whose key is below the target's current prefix range
        unsigned prefix[4] = {0, 0, 0, 0};
        const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
        for (int pass = 0; pass < 3 && !any_nan; ++pass) {
            const unsigned himask = pass == 0 ? 0u : ~((1u << (shifts[pass] + bits[pass])) - 1u);
            // distinct prefixes -> histograms
            unsigned pq[4] = {0, 0, 0, 0};
            int which[4], nq = 0;
            for (int q = 0; q < 4; ++q) {
                int f = -1;
                for (int k = 0; k < nq; ++k)
                    if (pq[k] == (prefix[q] & himask)) f = k;
                if (f < 0) {
                    pq[nq] = prefix[q] & himask;
                    f = nq++;
                }
                which[q] = f;
            }
            HIP_TRY(hipMemsetAsync(d_cnt, 0, (4 * 2048 + 1) * sizeof(unsigned long long), st));
            hipLaunchKernelGGL(icv::k_key_hist, dim3((unsigned)grid), dim3(256), 0, st, c, m, shifts[pass], bits[pass], himask,
                               pq[0], pq[1], pq[2], pq[3], nq, d_cnt);
            HIP_TRY(hipMemcpyAsync(h.data(), d_cnt, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (h[4 * 2048]) {
                any_nan = true;
                break;
            }
            for (int q = 0; q < 4; ++q) {
                const unsigned long long* hq = h.data() + (size_t)which[q] * 2048;
                unsigned long long run = below[q];
                unsigned d = 0;
                const unsigned nd = 1u << bits[pass];
                for (; d + 1 < nd; ++d) {
                    if (run + hq[d] > target[q]) break;
                    run += hq[d];
                }
                below[q] = run;
                prefix[q] |= d << shifts[pass];
            }
        }
        for (int q = 0; q < 4; ++q) lo[q] = prefix[q];
    }
    if (any_nan) {
        *h_iqr = std::nan("");
        return ICV_OK;
    }
    double qv[2];
    for (int q = 0; q < 2; ++q) {
        const double a = (double)icv::from_ordered_key32(lo[2 * q]), b = (double)icv::from_ordered_key32(lo[2 * q + 1]);
        const double t = pos[q] - std::floor(pos[q]);
        const double d = b - a;
        qv[q] = t >= 0.5 ? b - d * (1.0 - t) : a + d * t;  // numpy's _lerp
    }
    *h_iqr = qv[1] - qv[0];
    return ICV_OK;
}

}  // extern "C"

namespace {
// centred float32 copy of the points (distances are translation invariant) + float64 squared norms
struct CentredPoints {
    float* z = nullptr;
    double* work = nullptr;  // partial[n_slabs * d] | mean[d] | norm[n]
    double* norm = nullptr;
    int kz = 0;
    ~CentredPoints() {
        (void)hipFree(z);
        (void)hipFree(work);
    }
};
int centre_points(hipStream_t st, const float* x, int64_t n, int32_t d, int64_t ld, CentredPoints& c) {
    c.kz = icv::round_up(d, icv::GK);
    const int n_slabs = (int)std::min<int64_t>(256, (n + 63) / 64);
    HIP_TRY(hipMalloc((void**)&c.z, (size_t)n * c.kz * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c.work, ((size_t)n_slabs * d + d + n) * sizeof(double)));
    double *partial = c.work, *mean = c.work + (size_t)n_slabs * d;
    c.norm = mean + d;
    const unsigned gd = (unsigned)((d + 255) / 256);
    hipLaunchKernelGGL(icv::k_colsum_slabs, dim3(gd, n_slabs), dim3(256), 0, st, x, n, d, ld, n_slabs, partial);
    hipLaunchKernelGGL(icv::k_colmean_finish, dim3(gd), dim3(256), 0, st, partial, n, d, n_slabs, mean);
    hipLaunchKernelGGL(icv::k_center_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x, n, d, ld, mean, c.z, c.kz,
                       c.norm);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}
}  // namespace

extern "C" {

int icv_pairwise_sqeuclidean(const float* x, int64_t n, int32_t d, int64_t ld, int64_t row_begin, int64_t row_end,
                             float* out, int64_t ldo, void* stream) {
    if (!x || !out || n < 1 || d < 1 || ld < d || ldo < n || row_begin < 0 || row_end > n || row_begin > row_end)
        return fail(ICV_ERR_INVALID, "bad pairwise arguments");
    if (row_begin == row_end) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    CentredPoints c;
    if (int rc = centre_points(st, x, n, d, ld, c)) return rc;
    GramWork gw;
    if (row_begin == 0 && row_end == n) {
        if (int rc = launch_gram<true, true>(st, gw, c.z, c.kz, c.norm, gram_supers_sym(n, ldo), n, n, out, ldo, out, ldo))
            return rc;
    } else {  // a row block against all columns
        const int64_t ss = (int64_t)icv::GT * icv::GSUPER;
        std::vector<icv::GramSuper> v;
        for (int64_t r0 = row_begin; r0 < row_end; r0 += ss)
            for (int64_t c0 = 0; c0 < n; c0 += ss) v.push_back({(int)r0, (int)c0, (r0 - row_begin) * ldo + c0, 0});
        if (int rc = launch_gram<true, false>(st, gw, c.z, c.kz, c.norm, v, row_end, n, out, ldo, out, ldo)) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));  // the temporaries are freed on return
    return ICV_OK;
}

int icv_pairwise_sqeuclidean_tiles(const float* x, int64_t n, int32_t d, int64_t ld, int32_t n_tiles,
                                   const int32_t* h_row0, const int32_t* h_col0, const int64_t* h_dir_off,
                                   const int64_t* h_mir_off, float* dir, int64_t ld_dir, float* mir, int64_t ld_mir,
                                   void* stream) {
    static_assert(ICV_SUPER_ROWS == icv::GT * icv::GSUPER, "super-tile size");
    if (!x || n < 1 || d < 1 || ld < d || n_tiles < 0) return fail(ICV_ERR_INVALID, "bad pairwise tile arguments");
    if (n_tiles == 0) return ICV_OK;
    if (!h_row0 || !h_col0 || !h_dir_off || !h_mir_off || !dir || !mir)
        return fail(ICV_ERR_INVALID, "bad pairwise tile arguments");
    std::vector<icv::GramSuper> v((size_t)n_tiles);
    for (int32_t k = 0; k < n_tiles; ++k) {
        if (h_row0[k] < 0 || h_row0[k] % ICV_SUPER_ROWS || h_col0[k] % ICV_SUPER_ROWS || h_col0[k] < h_row0[k] ||
            h_col0[k] >= n)
            return fail(ICV_ERR_INVALID, "pairwise tiles: origins must be multiples of ICV_SUPER_ROWS on or above the diagonal");
        v[(size_t)k] = {h_row0[k], h_col0[k], h_dir_off[k], h_mir_off[k]};
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    CentredPoints c;
    if (int rc = centre_points(st, x, n, d, ld, c)) return rc;
    GramWork gw;
    if (int rc = launch_gram<true, true>(st, gw, c.z, c.kz, c.norm, v, n, n, dir, ld_dir, mir, ld_mir)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return ICV_OK;
}

}  // extern "C"

// ---- Ward rounds: device bookkeeping shared by the one-call and the step-wise entry points ---------------
// Two column layouts (icv_ward.hpp / icv_ward_strip.hpp): "strip" when the row stride leaves n / 2 spare columns
// (merged clusters get new, consecutive columns: dense strip updates), "in place" otherwise.
struct icv_ward_s {
    int64_t n = 0, ld = 0;
    int cap = 0;         // usable columns of a row (strip layout)
    bool strip = false;
    char* buf = nullptr;
    int *live, *cstate, *size_old, *size_new, *nn, *log_i, *log_j, *log_size, *act, *pslot, *vrow, *ulist;
    float *pair_d, *dmin, *log_d;
    unsigned char *alive, *qmask;
    int4 *mdesc, *mpos;
    icv::WardStripCounts* counts;
    icv::WardPos pos{};
    int *sr_local = nullptr, *sr_global = nullptr;  // device copies (sharded matrices)
    icv::WardMap map{nullptr, nullptr, 10, 0};
    icv::WardStripCounts h{0, 0, 0, 0, 0, 0, 0, 0};  // after the last icv_ward_pairs
    int rounds = 0;
    ~icv_ward_s() { (void)hipFree(buf); }
    bool vec_ok(const void* D, int64_t ld_) const { return (ld_ % 4 == 0) && ((reinterpret_cast<uintptr_t>(D) & 15) == 0); }
    // stream over all columns, or gather over the live list?
    bool dense(const void* D, int64_t ld_, int width) const { return vec_ok(D, ld_) && (int64_t)h.n_live * 4 >= width; }
    int merged_begin() const { return h.n_merges - h.n_pairs; }
};

namespace {
int ward_create(int64_t n, const int32_t* sr_local, int32_t n_super, int32_t super_shift, int64_t ld, bool spare,
                hipStream_t st, icv_ward_s** out) {
    if (n < 2 || n > 0x7fffffff / 2 || !out || ld < n) return fail(ICV_ERR_INVALID, "bad ward arguments");
    if (spare && (ld % 4 != 0 || ld < n + (n + 1) / 2))
        return fail(ICV_ERR_INVALID, "ward: spare columns need a row stride that is a multiple of 4 and >= n + (n + 1) / 2");
    if (sr_local && (n_super < 1 || super_shift < 2 || super_shift > 20 || ((int64_t)n_super << super_shift) < n))
        return fail(ICV_ERR_INVALID, "bad ward storage map");
    std::unique_ptr<icv_ward_s> w(new icv_ward_s);
    w->n = n;
    w->ld = ld;
    // the caller says whether columns [n, ld) of every row are the rounds' to use (never inferred from the stride: a
    // column slice of a wider buffer has a large stride too); ICV_WARD_IN_PLACE: developer knob, the other layout
    w->strip = spare && !knobs().ward_in_place;
    w->cap = w->strip ? (int)std::min<int64_t>(ld, 2 * n) : (int)n;
    const size_t arr = ((size_t)n * 4 + 255) / 256 * 256;
    const size_t parr = ((size_t)w->cap * 4 + 255) / 256 * 256;  // arrays indexed by column position
    const size_t map_bytes = sr_local ? ((size_t)n_super * 8 + 255) / 256 * 256 : 0;
    // 16 slot-indexed arrays + 2 x 4 (merge descriptors) + slot_pos; 6 position-indexed arrays; counters; storage map
    HIP_TRY(hipMalloc((void**)&w->buf, arr * 25 + parr * 6 + 512 + map_bytes));
    char* b = w->buf;
    auto take = [&](size_t bytes) {
        char* p = b;
        b += bytes;
        return p;
    };
    w->live = (int*)take(arr);
    w->cstate = (int*)take(arr);
    w->pair_d = (float*)take(arr);
    w->size_old = (int*)take(arr);
    w->size_new = (int*)take(arr);
    w->nn = (int*)take(arr);
    w->dmin = (float*)take(arr);
    w->log_i = (int*)take(arr);
    w->log_j = (int*)take(arr);
    w->log_d = (float*)take(arr);
    w->log_size = (int*)take(arr);
    w->alive = (unsigned char*)take(arr);
    w->act = (int*)take(arr);
    w->pslot = (int*)take(arr);
    w->vrow = (int*)take(arr);
    w->ulist = (int*)take(arr);
    w->mdesc = (int4*)take(arr * 4);  // n x 16 bytes: one packed descriptor per merge
    w->mpos = (int4*)take(arr * 4);
    w->pos.slot_pos = (int*)take(arr);
    w->qmask = (unsigned char*)take(parr);
    w->pos.pos_slot = (int*)take(parr);
    w->pos.palive = (unsigned char*)take(parr);
    w->pos.pstate = (int*)take(parr);
    w->pos.psize = (int*)take(parr);
    w->pos.newpos = (int*)take(parr);
    w->counts = (icv::WardStripCounts*)take(256);
    if (sr_local) {
        // local super-row index per global super-row and its inverse
        std::vector<int> inv;
        for (int g = 0; g < n_super; ++g)
            if (sr_local[g] >= 0) {
                if ((size_t)sr_local[g] >= inv.size()) inv.resize((size_t)sr_local[g] + 1, -1);
                inv[(size_t)sr_local[g]] = g;
            }
        for (int v : inv)
            if (v < 0) return fail(ICV_ERR_INVALID, "ward storage map: local super-rows must be 0 .. k-1");
        w->sr_local = (int*)b;
        w->sr_global = w->sr_local + n_super;
        HIP_TRY(hipMemcpyAsync(w->sr_local, sr_local, (size_t)n_super * 4, hipMemcpyHostToDevice, st));
        if (!inv.empty())
            HIP_TRY(hipMemcpyAsync(w->sr_global, inv.data(), inv.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        int64_t n_local = 0;
        for (size_t l = 0; l < inv.size(); ++l) {
            const int64_t r0 = (int64_t)inv[l] << super_shift;
            n_local = std::max<int64_t>(n_local, ((int64_t)l << super_shift) + std::min<int64_t>(n - r0, (int64_t)1 << super_shift));
        }
        w->map = icv::WardMap{w->sr_local, w->sr_global, super_shift, (int)n_local};
    } else {
        w->map = icv::WardMap{nullptr, nullptr, 10, (int)n};
    }
    // both layouts start from the same state: width n, every row to be searched
    hipLaunchKernelGGL(icv::k_ward_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (int)n, w->live,
                       w->cstate, w->qmask, w->size_old, w->size_new, w->alive, w->act,
                       reinterpret_cast<icv::WardCounts*>(w->counts));
    if (w->strip)
        hipLaunchKernelGGL(icv::k_ward_init_s, dim3((unsigned)((w->cap + 255) / 256)), dim3(256), 0, st, (int)n, w->cap,
                           w->pos, w->ulist, w->qmask, w->counts);
    HIP_TRY(hipGetLastError());
    w->h = icv::WardStripCounts{(int)n, 0, 0, (int)n, (int)n, (int)n, 0, (int)n};
    *out = w.release();
    return ICV_OK;
}

int ward_check_ld(const icv_ward_s* w, int64_t ld, const void* D = nullptr) {
    if (ld != w->ld) return fail(ICV_ERR_INVALID, "ward: the row stride differs from the one given to icv_ward_create");
    if (w->strip && D && (reinterpret_cast<uintptr_t>(D) & 15))
        return fail(ICV_ERR_INVALID, "ward: a matrix with spare columns must be 16-byte aligned");
    return ICV_OK;
}

int ward_merge(icv_ward_s* w, float* D, int64_t ld, const float* stage, int64_t ld_stage, const int32_t* h_pslot,
               bool scatter, hipStream_t st) {
    if (int rc = ward_check_ld(w, ld, D)) return rc;
    if (w->h.n_pairs < 1) return ICV_OK;
    const int mb = w->merged_begin();
    if (h_pslot) HIP_TRY(hipMemcpyAsync(w->pslot, h_pslot, (size_t)w->h.n_pairs * 4, hipMemcpyHostToDevice, st));
    const icv::WardPairView V{w->mdesc + mb, w->log_d + mb, h_pslot ? w->pslot : nullptr, stage, ld_stage};
    const bool stage_ok = !h_pslot || ((ld_stage % 4 == 0) && ((reinterpret_cast<uintptr_t>(stage) & 15) == 0));
    const dim3 grid((unsigned)w->h.n_pairs), block(256);
    if (w->strip) {
        if (w->dense(D, ld, w->h.width_prev) && stage_ok)
            hipLaunchKernelGGL(icv::k_ward_merge_s<true>, grid, block, 0, st, D, ld, w->h.width_prev, w->live, w->h.n_live,
                               w->cstate, V, w->mpos + mb, w->h.n_pairs, w->pos, w->map, w->nn, w->dmin);
        else
            hipLaunchKernelGGL(icv::k_ward_merge_s<false>, grid, block, 0, st, D, ld, w->h.width_prev, w->live, w->h.n_live,
                               w->cstate, V, w->mpos + mb, w->h.n_pairs, w->pos, w->map, w->nn, w->dmin);
        if (scatter && w->h.n_unmerged > 0) {  // one GPU: the strip update straight from the new rows
            const dim3 g2((unsigned)((w->h.n_pairs + 63) / 64), (unsigned)((w->h.n_unmerged + 63) / 64));
            hipLaunchKernelGGL(icv::k_ward_push, g2, block, 0, st, D, ld, w->h.width_prev, w->mdesc + mb, w->h.n_pairs,
                               w->ulist, w->h.n_unmerged, w->pos.slot_pos);
        }
    } else if (w->dense(D, ld, (int)w->n) && stage_ok) {
        hipLaunchKernelGGL(icv::k_ward_merge<true>, grid, block, 0, st, D, ld, (int)w->n, w->live, w->h.n_live, w->cstate,
                           V, w->pair_d, w->size_old, w->size_new, w->map, scatter, w->nn, w->dmin);
    } else {
        hipLaunchKernelGGL(icv::k_ward_merge<false>, grid, block, 0, st, D, ld, (int)w->n, w->live, w->h.n_live, w->cstate,
                           V, w->pair_d, w->size_old, w->size_new, w->map, scatter, w->nn, w->dmin);
    }
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int ward_scan(icv_ward_s* w, const float* D, int64_t ld, hipStream_t st) {
    if (int rc = ward_check_ld(w, ld, D)) return rc;
    if (w->h.n_act < 1) return ICV_OK;
    const dim3 grid((unsigned)w->h.n_act), block(256);
    if (w->strip) {
        if (w->dense(D, ld, w->h.width))
            hipLaunchKernelGGL(icv::k_ward_scan_s<true>, grid, block, 0, st, D, ld, w->h.width, w->act, w->live,
                               w->h.n_live, w->qmask, w->pos, w->map, w->nn, w->dmin);
        else
            hipLaunchKernelGGL(icv::k_ward_scan_s<false>, grid, block, 0, st, D, ld, w->h.width, w->act, w->live,
                               w->h.n_live, w->qmask, w->pos, w->map, w->nn, w->dmin);
    } else if (w->dense(D, ld, (int)w->n)) {
        hipLaunchKernelGGL(icv::k_ward_scan<true>, grid, block, 0, st, D, ld, (int)w->n, w->act, w->live, w->h.n_live,
                           w->qmask, w->map, w->nn, w->dmin);
    } else {
        hipLaunchKernelGGL(icv::k_ward_scan<false>, grid, block, 0, st, D, ld, (int)w->n, w->act, w->live, w->h.n_live,
                           w->qmask, w->map, w->nn, w->dmin);
    }
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int ward_read_counts(icv_ward_s* w, hipStream_t st) {
    if (w->strip) {
        HIP_TRY(hipMemcpyAsync(&w->h, w->counts, sizeof(w->h), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        icv::WardCounts c;
        HIP_TRY(hipMemcpyAsync(&c, w->counts, sizeof(c), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        w->h = icv::WardStripCounts{c.n_live, c.n_merges, c.n_pairs, c.n_act, (int)w->n, (int)w->n, 0, 0};
    }
    return ICV_OK;
}

// in-place compaction of the alive columns of the alive local rows (strip layout)
int ward_compact(icv_ward_s* w, float* D, int64_t ld, hipStream_t st) {
    if (int rc = ward_check_ld(w, ld, D)) return rc;
    if (!w->strip) return ICV_OK;
    hipLaunchKernelGGL(icv::k_ward_compact_map, dim3(1), dim3(1024), 0, st, w->cap, w->pos, w->qmask, w->counts);
    if (w->h.n_live > 0 && w->map.n_local > 0)
        hipLaunchKernelGGL(icv::k_ward_compact_rows, dim3((unsigned)w->h.n_live), dim3(256), 0, st, D, ld, w->h.width,
                           w->live, w->pos.newpos, w->map);
    HIP_TRY(hipGetLastError());
    w->h.width_prev = w->h.width;
    w->h.width = w->h.n_live;
    w->h.need_compact = 0;
    return ICV_OK;
}

// reciprocal pairs of the round; strip layout: compacts first when fewer than half of the positions are alive,
// and again (then it must fit) when the round's new columns would not fit the spare region
int ward_pairs(icv_ward_s* w, float* D, int64_t ld, bool all_active, hipStream_t st) {
    if (!w->strip) {
        hipLaunchKernelGGL(icv::k_ward_pairs, dim3(1), dim3(1024), 0, st, (int)w->n, w->live, w->cstate, w->qmask, w->mdesc,
                           w->pair_d, w->size_old, w->size_new, w->alive, w->nn, w->dmin, w->log_i, w->log_j, w->log_d,
                           w->log_size, w->act, all_active ? 1 : 0, reinterpret_cast<icv::WardCounts*>(w->counts));
        HIP_TRY(hipGetLastError());
        ++w->rounds;
        return ward_read_counts(w, st);
    }
    {
        // developer knob: compaction threshold (positions per alive column), default 2
        const double kx = knobs().ward_compact_x;
        const double thr = kx >= 1.05 ? kx : 2.0;
        if ((double)w->h.width > thr * (double)w->h.n_live && w->h.width > 4096)
            if (int rc = ward_compact(w, D, ld, st)) return rc;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int span = std::max<int>((int)w->n, w->cap);
        hipLaunchKernelGGL(icv::k_ward_prep_s, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st, (int)w->n, w->cstate,
                           w->size_old, w->size_new, w->alive, w->pos, w->counts);
        hipLaunchKernelGGL(icv::k_ward_pairs_s, dim3(1), dim3(1024), 0, st, (int)w->n, w->cap, w->live, w->cstate,
                           w->mdesc, w->mpos, w->pair_d, w->size_old, w->size_new, w->alive, w->nn, w->dmin, w->log_i,
                           w->log_j, w->log_d, w->log_size, w->act, w->ulist, all_active ? 1 : 0, w->pos, w->counts);
        hipLaunchKernelGGL(icv::k_ward_qmask_s, dim3((unsigned)(((w->cap + 3) / 4 + 255) / 256)), dim3(256), 0, st, w->cap,
                           w->pos.palive, w->qmask);
        HIP_TRY(hipGetLastError());
        const int width_before = w->h.width;
        if (int rc = ward_read_counts(w, st)) return rc;
        if (!w->h.need_compact) {
            ++w->rounds;
            return ICV_OK;
        }
        if (attempt == 1) break;
        // nothing was committed: the counters other than need_compact are those of the previous round
        w->h.width = width_before;
        if (int rc = ward_compact(w, D, ld, st)) return rc;
    }
    return fail(ICV_ERR_UNSUPPORTED, "ward_pairs: the spare columns do not hold one round");
}

// merge log -> scipy linkage matrix: monotone heights, stable sort, union-find relabelling
int ward_finish(icv_ward_s* w, double* h_linkage) {
    const int64_t n = w->n;
    const size_t m = (size_t)n - 1;
    if ((size_t)w->h.n_merges != m) return fail(ICV_ERR_INVALID, "ward_finish: the rounds are not complete");
    std::vector<int> li(m), lj(m), ls(m);
    std::vector<float> ldq(m);
    HIP_TRY(hipMemcpy(li.data(), w->log_i, m * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lj.data(), w->log_j, m * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ls.data(), w->log_size, m * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ldq.data(), w->log_d, m * 4, hipMemcpyDeviceToHost));
    std::vector<double> height(m), slot_h((size_t)n, 0.0);
    for (size_t p = 0; p < m; ++p) {
        double h = std::sqrt((double)ldq[p]);
        h = std::max(h, std::max(slot_h[li[p]], slot_h[lj[p]]));  // a parent never sorts before its children
        height[p] = h;
        slot_h[li[p]] = h;
    }
    std::vector<size_t> order(m);
    for (size_t p = 0; p < m; ++p) order[p] = p;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return height[a] < height[b]; });
    std::vector<int64_t> cluster((size_t)n);  // current scipy id of the cluster kept in each slot
    for (int64_t s = 0; s < n; ++s) cluster[s] = s;
    for (size_t q = 0; q < m; ++q) {
        const size_t p = order[q];
        const int64_t a = cluster[li[p]], b = cluster[lj[p]];
        h_linkage[4 * q + 0] = (double)std::min(a, b);
        h_linkage[4 * q + 1] = (double)std::max(a, b);
        h_linkage[4 * q + 2] = height[p];
        h_linkage[4 * q + 3] = (double)ls[p];
        cluster[li[p]] = n + (int64_t)q;
    }
    return ICV_OK;
}
}  // namespace

extern "C" {

int icv_ward_linkage(float* dist_sq, int64_t n, int64_t ld, int32_t spare_columns, double* h_linkage,
                     int32_t* h_rounds, void* stream) {
    if (!dist_sq || !h_linkage || n < 1 || ld < n || n > 0x7fffffff / 2)
        return fail(ICV_ERR_INVALID, "bad ward_linkage arguments");
    if (h_rounds) *h_rounds = 0;
    if (n == 1) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    icv_ward_s* raw = nullptr;
    if (int rc = ward_create(n, nullptr, 0, 0, ld, spare_columns != 0, st, &raw)) return rc;
    std::unique_ptr<icv_ward_s> w(raw);
    bool retry = false;
    while (w->h.n_live > 1) {
        if (int rc = ward_merge(w.get(), dist_sq, ld, nullptr, 0, nullptr, true, st)) return rc;
        if (int rc = ward_scan(w.get(), dist_sq, ld, st)) return rc;
        if (int rc = ward_pairs(w.get(), dist_sq, ld, false, st)) return rc;
        if (w->h.n_pairs < 1) {
            // No reciprocal pair: possible only when cached neighbours of TIED distances point in a cycle (fresh
            // neighbours under one total order always contain a reciprocal pair).  List every live row, search
            // them all again, and give up if that does not help (non-finite distances).
            if (retry) return fail(ICV_ERR_INVALID, "ward_linkage: distances are not finite");
            retry = true;
            if (int rc = ward_pairs(w.get(), dist_sq, ld, true, st)) return rc;
            --w->rounds;  // bookkeeping only, not a round
            if (w->h.n_pairs > 0) retry = false;
        } else {
            retry = false;
        }
    }
    if (h_rounds) *h_rounds = w->rounds;
    return ward_finish(w.get(), h_linkage);
}

int icv_ward_create(int64_t n, const int32_t* sr_local, int32_t n_super, int32_t super_shift, int64_t ld,
                    int32_t spare_columns,
                    icv_ward_t* out, void* stream) {
    return ward_create(n, sr_local, n_super, super_shift, ld, spare_columns != 0, static_cast<hipStream_t>(stream), out);
}
void icv_ward_destroy(icv_ward_t w) { delete w; }

int icv_ward_merge(icv_ward_t w, float* d_local, int64_t ld, const float* stage, int64_t ld_stage,
                   const int32_t* h_pslot, int32_t scatter, void* stream) {
    if (!w || !d_local) return fail(ICV_ERR_INVALID, "bad ward_merge arguments");
    return ward_merge(w, d_local, ld, stage, ld_stage, h_pslot, scatter != 0, static_cast<hipStream_t>(stream));
}

int icv_ward_gather(icv_ward_t w, const float* d_local, int64_t ld, const int64_t* d_rows, int32_t n_rows,
                    const int32_t* d_slots, int32_t n_slots, float* out, int64_t ldo, void* stream) {
    if (!w) return fail(ICV_ERR_INVALID, "bad ward_gather arguments");
    if (int rc = ward_check_ld(w, ld)) return rc;
    if (n_rows < 1 || n_slots < 1) return ICV_OK;
    if (!d_local || !d_rows || !d_slots || !out || ldo < n_slots) return fail(ICV_ERR_INVALID, "bad ward_gather arguments");
    hipLaunchKernelGGL(icv::k_ward_gather, dim3((unsigned)((n_slots + 255) / 256), (unsigned)n_rows), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_local, ld, d_rows, d_slots, n_slots,
                       w->strip ? w->pos.slot_pos : nullptr, w->strip ? w->h.width : (int)w->n, out, ldo);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_ward_scatter(icv_ward_t w, float* d_local, int64_t ld, const float* v, int64_t ldv, const int32_t* h_vrow_p,
                     int32_t n_v, void* stream) {
    if (!w || !d_local || n_v < 0 || n_v > w->n) return fail(ICV_ERR_INVALID, "bad ward_scatter arguments");
    if (int rc = ward_check_ld(w, ld)) return rc;
    if (n_v == 0 || w->map.n_local < 1) return ICV_OK;
    if (!v || !h_vrow_p || ldv < w->map.n_local || n_v != w->h.n_pairs)
        return fail(ICV_ERR_INVALID, "bad ward_scatter arguments (one row per merge of the round)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int mb = w->merged_begin();
    if (w->strip) {
        std::vector<int> inv((size_t)n_v, -1);  // merge index -> row of v
        for (int q = 0; q < n_v; ++q) {
            if (h_vrow_p[q] < 0 || h_vrow_p[q] >= n_v || inv[(size_t)h_vrow_p[q]] >= 0)
                return fail(ICV_ERR_INVALID, "ward_scatter: h_vrow_p must be a permutation of the round's merges");
            inv[(size_t)h_vrow_p[q]] = q;
        }
        HIP_TRY(hipMemcpyAsync(w->vrow, inv.data(), (size_t)n_v * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));  // inv is a temporary
        const dim3 grid((unsigned)((n_v + 63) / 64), (unsigned)((w->map.n_local + 63) / 64));
        hipLaunchKernelGGL(icv::k_ward_scatter_s, grid, dim3(256), 0, st, d_local, ld, w->h.width_prev, v, ldv, w->vrow, n_v,
                           w->cstate, w->map);
    } else {
        std::vector<int> slots((size_t)n_v);
        std::vector<int> li((size_t)n_v);
        HIP_TRY(hipMemcpyAsync(li.data(), w->log_i + mb, (size_t)n_v * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int q = 0; q < n_v; ++q) {
            if (h_vrow_p[q] < 0 || h_vrow_p[q] >= n_v) return fail(ICV_ERR_INVALID, "ward_scatter: bad merge index");
            slots[(size_t)q] = li[(size_t)h_vrow_p[q]];
        }
        HIP_TRY(hipMemcpyAsync(w->vrow, slots.data(), (size_t)n_v * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        hipLaunchKernelGGL(icv::k_ward_scatter, dim3((unsigned)((w->map.n_local + 255) / 256), (unsigned)n_v), dim3(256), 0,
                           st, d_local, ld, v, ldv, w->vrow, w->cstate, w->map);
    }
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_ward_scan(icv_ward_t w, const float* d_local, int64_t ld, void* stream) {
    if (!w || !d_local) return fail(ICV_ERR_INVALID, "bad ward_scan arguments");
    return ward_scan(w, d_local, ld, static_cast<hipStream_t>(stream));
}

int icv_ward_pack_nn(icv_ward_t w, int32_t* d_nn, float* d_dmin, void* stream) {
    if (!w || !d_nn || !d_dmin) return fail(ICV_ERR_INVALID, "bad ward_pack_nn arguments");
    const int k = w->h.n_pairs + w->h.n_act;
    if (k < 1) return ICV_OK;
    hipLaunchKernelGGL(icv::k_ward_pack_nn, dim3((unsigned)((k + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w->mdesc + w->merged_begin(), w->h.n_pairs, w->act, w->h.n_act,
                       w->nn, w->dmin, w->map, d_nn, d_dmin);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_ward_unpack_nn(icv_ward_t w, const int32_t* d_nn, const float* d_dmin, void* stream) {
    if (!w || !d_nn || !d_dmin) return fail(ICV_ERR_INVALID, "bad ward_unpack_nn arguments");
    const int k = w->h.n_pairs + w->h.n_act;
    if (k < 1) return ICV_OK;
    hipLaunchKernelGGL(icv::k_ward_unpack_nn, dim3((unsigned)((k + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w->mdesc + w->merged_begin(), w->h.n_pairs, w->act, w->h.n_act,
                       d_nn, d_dmin, w->nn, w->dmin);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_ward_pairs(icv_ward_t w, float* d_local, int64_t ld, int32_t all_active, int32_t* h_counts, void* stream) {
    if (!w || !h_counts || (!d_local && w->map.n_local > 0)) return fail(ICV_ERR_INVALID, "bad ward_pairs arguments");
    if (int rc = ward_check_ld(w, ld)) return rc;
    if (int rc = ward_pairs(w, d_local, ld, all_active != 0, static_cast<hipStream_t>(stream))) return rc;
    h_counts[0] = w->h.n_live;
    h_counts[1] = w->h.n_merges;
    h_counts[2] = w->h.n_pairs;
    h_counts[3] = w->h.n_act;
    return ICV_OK;
}

int icv_ward_round_pairs(icv_ward_t w, int32_t* h_i, int32_t* h_j) {
    if (!w || !h_i || !h_j) return fail(ICV_ERR_INVALID, "bad ward_round_pairs arguments");
    if (w->h.n_pairs < 1) return ICV_OK;
    HIP_TRY(hipMemcpy(h_i, w->log_i + w->merged_begin(), (size_t)w->h.n_pairs * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_j, w->log_j + w->merged_begin(), (size_t)w->h.n_pairs * 4, hipMemcpyDeviceToHost));
    return ICV_OK;
}

int icv_ward_finish(icv_ward_t w, double* h_linkage, int32_t* h_rounds) {
    if (!w || !h_linkage) return fail(ICV_ERR_INVALID, "bad ward_finish arguments");
    if (h_rounds) *h_rounds = w->rounds;
    return ward_finish(w, h_linkage);
}

int icv_row_abs_sum(const float* x, int64_t n_rows, int32_t n_cols, int64_t ld, double* row_sum, void* stream) {
    if (!x || !row_sum || n_cols < 0 || ld < n_cols) return fail(ICV_ERR_INVALID, "bad row_abs_sum arguments");
    if (n_rows < 1) return ICV_OK;
    hipLaunchKernelGGL(icv::k_row_abs_sum, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n_rows, n_cols, ld, row_sum);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_csr_row_abs_sum(const void* data, int32_t dtype, const int64_t* indptr, int64_t n_rows, double* row_sum,
                        void* stream) {
    if (!indptr || !row_sum || (dtype != ICV_F32 && dtype != ICV_F64))
        return fail(ICV_ERR_INVALID, "bad csr_row_abs_sum arguments");
    if (n_rows < 1) return ICV_OK;
    dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ICV_F32)
        hipLaunchKernelGGL(icv::k_csr_row_abs_sum<float>, grid, block, 0, st, static_cast<const float*>(data), indptr,
                           n_rows, row_sum);
    else
        hipLaunchKernelGGL(icv::k_csr_row_abs_sum<double>, grid, block, 0, st, static_cast<const double*>(data), indptr,
                           n_rows, row_sum);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_csr_check(const int64_t* indptr, const int32_t* indices, int64_t n_rows, int32_t n_cols, int64_t capacity,
                  void* stream) {
    if (!indptr || n_rows < 0 || n_cols < 0 || capacity < 0 || (capacity > 0 && !indices))
        return fail(ICV_ERR_INVALID, "bad csr_check arguments");
    if (n_rows == 0) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    AsyncBuf flag_b;
    HIP_TRY(flag_b.alloc(sizeof(int), st));
    HIP_TRY(hipMemsetAsync(flag_b.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(icv::k_csr_check, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, indptr, indices, n_rows,
                       n_cols, capacity, flag_b.as<int>());
    HIP_TRY(hipGetLastError());
    int h_flag = 0;
    HIP_TRY(hipMemcpyAsync(&h_flag, flag_b.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_flag & icv::kCsrBadOffsets)
        return fail(ICV_ERR_INVALID, "device CSR: row offsets must be non-decreasing and inside the index / value buffers");
    if (h_flag & icv::kCsrBadColumn) return fail(ICV_ERR_INVALID, "device CSR: a column index is outside [0, n_cols)");
    if (h_flag & icv::kCsrUnsorted)
        return fail(ICV_ERR_INVALID, "device CSR: column indices must be ascending and unique within every row "
                                     "(scipy: sum_duplicates() / sort_indices() before the upload)");
    return ICV_OK;
}

int icv_csr_densify(const void* data, int32_t dtype, const int64_t* indptr, const int32_t* indices, const int64_t* rows,
                    int64_t n_sel, int32_t n_cols, float* out, int64_t ldo, void* stream) {
    if (!indptr || !out || n_sel < 0 || n_cols < 0 || ldo < n_cols || (dtype != ICV_F32 && dtype != ICV_F64))
        return fail(ICV_ERR_INVALID, "bad csr_densify arguments");
    if (n_sel == 0 || n_cols == 0) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)((n_sel - 1) * ldo + n_cols) * sizeof(float), st));
    dim3 grid((unsigned)((n_sel + 3) / 4)), block(256);
    if (dtype == ICV_F32)
        hipLaunchKernelGGL(icv::k_csr_densify<float>, grid, block, 0, st, static_cast<const float*>(data), indptr, indices,
                           rows, n_sel, out, ldo);
    else
        hipLaunchKernelGGL(icv::k_csr_densify<double>, grid, block, 0, st, static_cast<const double*>(data), indptr,
                           indices, rows, n_sel, out, ldo);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

}  // extern "C"

// ---- host-side packing of a mostly-zero dense matrix (the upload path of tl.infercnv for dense host input) ---------------
namespace {
template <typename T, typename B>
void host_count_rows(const T* x, int64_t r0, int64_t r1, int64_t n_cols, int64_t ld, int64_t* row_nnz) {
    for (int64_t r = r0; r < r1; ++r) {
        const B* b = reinterpret_cast<const B*>(x + r * ld);
        int64_t c = 0;
        for (int64_t j = 0; j < n_cols; ++j) c += b[j] != 0;  // bit pattern: NaN and -0.0 are stored entries
        row_nnz[r] = c;
    }
}
// elements [j, n_cols) of one row, entries from cursor k on (k < end while there is something to find): branch-free,
// every element is written at the cursor and the cursor moves on for stored ones; never writes at or past `end`
template <typename T, typename B>
inline void host_pack_tail(const T* xr, int64_t j, int64_t n_cols, int64_t k, int64_t end, int32_t* indices, T* values) {
    const B* b = reinterpret_cast<const B*>(xr);
    for (; j < n_cols && k < end; ++j) {
        indices[k] = (int32_t)j;
        values[k] = xr[j];
        k += b[j] != 0;
    }
}
template <typename T, typename B>
void host_pack_rows(const T* x, int64_t r0, int64_t r1, int64_t n_cols, int64_t ld, const int64_t* indptr,
                    int32_t* indices, T* values) {
    for (int64_t r = r0; r < r1; ++r) {
        const T* xr = x + r * ld;
        const int64_t k = indptr[r], end = indptr[r + 1];
        if (end - k == n_cols) {  // a full row: no test per element
            for (int64_t j = 0; j < n_cols; ++j) indices[k + j] = (int32_t)j, values[k + j] = xr[j];
            continue;
        }
        host_pack_tail<T, B>(xr, 0, n_cols, k, end, indices, values);
    }
}

// AVX-512 forms (chosen at run time: the library is built on one machine and runs on another).  Register compress +
// full-width store (the memory-destination compress is microcoded on some cores); a vector is only stored while 16 (8)
// slots of THIS row are left, the last entries of a row go through the scalar tail -- nothing is written outside
// [indptr[r], indptr[r + 1]), so rows can be packed by different threads.
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt")))
void host_count_rows_avx512_f32(const float* x, int64_t r0, int64_t r1, int64_t n_cols, int64_t ld, int64_t* row_nnz) {
    for (int64_t r = r0; r < r1; ++r) {
        const float* xr = x + r * ld;
        int64_t c = 0, j = 0;
        for (; j + 16 <= n_cols; j += 16) {
            const __m512i v = _mm512_loadu_si512((const void*)(xr + j));
            c += _mm_popcnt_u32((unsigned)_mm512_test_epi32_mask(v, v));
        }
        const uint32_t* b = reinterpret_cast<const uint32_t*>(xr);
        for (; j < n_cols; ++j) c += b[j] != 0;
        row_nnz[r] = c;
    }
}
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt")))
void host_count_rows_avx512_f64(const double* x, int64_t r0, int64_t r1, int64_t n_cols, int64_t ld, int64_t* row_nnz) {
    for (int64_t r = r0; r < r1; ++r) {
        const double* xr = x + r * ld;
        int64_t c = 0, j = 0;
        for (; j + 8 <= n_cols; j += 8) {
            const __m512i v = _mm512_loadu_si512((const void*)(xr + j));
            c += _mm_popcnt_u32((unsigned)_mm512_test_epi64_mask(v, v));
        }
        const uint64_t* b = reinterpret_cast<const uint64_t*>(xr);
        for (; j < n_cols; ++j) c += b[j] != 0;
        row_nnz[r] = c;
    }
}
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt")))
void host_pack_rows_avx512_f32(const float* x, int64_t r0, int64_t r1, int64_t n_cols, int64_t ld, const int64_t* indptr,
                               int32_t* indices, float* values) {
    const __m512i iota = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    for (int64_t r = r0; r < r1; ++r) {
        const float* xr = x + r * ld;
        int64_t k = indptr[r];
        const int64_t end = indptr[r + 1];
        int64_t j = 0;
        for (; j + 16 <= n_cols && k + 16 <= end; j += 16) {
            const __m512i v = _mm512_loadu_si512((const void*)(xr + j));
            const __mmask16 m = _mm512_test_epi32_mask(v, v);
            _mm512_storeu_si512((void*)(values + k), _mm512_maskz_compress_epi32(m, v));
            _mm512_storeu_si512((void*)(indices + k),
                                _mm512_maskz_compress_epi32(m, _mm512_add_epi32(iota, _mm512_set1_epi32((int)j))));
            k += _mm_popcnt_u32((unsigned)m);
        }
        host_pack_tail<float, uint32_t>(xr, j, n_cols, k, end, indices, values);
    }
}
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt")))
void host_pack_rows_avx512_f64(const double* x, int64_t r0, int64_t r1, int64_t n_cols, int64_t ld, const int64_t* indptr,
                               int32_t* indices, double* values) {
    const __m256i iota = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    for (int64_t r = r0; r < r1; ++r) {
        const double* xr = x + r * ld;
        int64_t k = indptr[r];
        const int64_t end = indptr[r + 1];
        int64_t j = 0;
        for (; j + 8 <= n_cols && k + 8 <= end; j += 8) {
            const __m512i v = _mm512_loadu_si512((const void*)(xr + j));
            const __mmask8 m = _mm512_test_epi64_mask(v, v);
            _mm512_storeu_si512((void*)(values + k), _mm512_maskz_compress_epi64(m, v));
            _mm256_storeu_si256((__m256i*)(indices + k),
                                _mm256_maskz_compress_epi32(m, _mm256_add_epi32(iota, _mm256_set1_epi32((int)j))));
            k += _mm_popcnt_u32((unsigned)m);
        }
        host_pack_tail<double, uint64_t>(xr, j, n_cols, k, end, indices, values);
    }
}
// one row into a staging area with 16 slots of slack behind the row's worst case: no bound to respect, returns the count
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt")))
int64_t host_pack_row_avx512_f32(const float* xr, int64_t n_cols, int32_t* indices, float* values) {
    const __m512i iota = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    int64_t k = 0, j = 0;
    for (; j + 16 <= n_cols; j += 16) {
        const __m512i v = _mm512_loadu_si512((const void*)(xr + j));
        const __mmask16 m = _mm512_test_epi32_mask(v, v);
        _mm512_storeu_si512((void*)(values + k), _mm512_maskz_compress_epi32(m, v));
        _mm512_storeu_si512((void*)(indices + k),
                            _mm512_maskz_compress_epi32(m, _mm512_add_epi32(iota, _mm512_set1_epi32((int)j))));
        k += _mm_popcnt_u32((unsigned)m);
    }
    const uint32_t* b = reinterpret_cast<const uint32_t*>(xr);
    for (; j < n_cols; ++j) {
        indices[k] = (int32_t)j;
        values[k] = xr[j];
        k += b[j] != 0;
    }
    return k;
}
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt")))
int64_t host_pack_row_avx512_f64(const double* xr, int64_t n_cols, int32_t* indices, double* values) {
    const __m256i iota = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    int64_t k = 0, j = 0;
    for (; j + 8 <= n_cols; j += 8) {
        const __m512i v = _mm512_loadu_si512((const void*)(xr + j));
        const __mmask8 m = _mm512_test_epi64_mask(v, v);
        _mm512_storeu_si512((void*)(values + k), _mm512_maskz_compress_epi64(m, v));
        _mm256_storeu_si256((__m256i*)(indices + k),
                            _mm256_maskz_compress_epi32(m, _mm256_add_epi32(iota, _mm256_set1_epi32((int)j))));
        k += _mm_popcnt_u32((unsigned)m);
    }
    const uint64_t* b = reinterpret_cast<const uint64_t*>(xr);
    for (; j < n_cols; ++j) {
        indices[k] = (int32_t)j;
        values[k] = xr[j];
        k += b[j] != 0;
    }
    return k;
}
bool host_has_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
                           __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("popcnt") &&
                           std::getenv("ICV_NO_AVX512") == nullptr;
    return ok;
}
#else
bool host_has_avx512() { return false; }
#endif
template <typename T, typename B>
int64_t host_pack_row_scalar(const T* xr, int64_t n_cols, int32_t* indices, T* values) {
    const B* b = reinterpret_cast<const B*>(xr);
    int64_t k = 0;
    for (int64_t j = 0; j < n_cols; ++j) {
        indices[k] = (int32_t)j;
        values[k] = xr[j];
        k += b[j] != 0;
    }
    return k;
}

// ONE pass over the input: threads claim blocks of kFusedRows rows in order, pack a block into their own staging area
// (cache resident at log-count densities), take the block's place in the output through a ticket that is handed on in
// block order (a running entry count), and copy the block there.  The input is read once; the two-pass form (count,
// prefix sums, pack) reads it twice, and the host memory system is what the upload of a 16 GB matrix waits for.
constexpr int64_t kFusedRows = 16;

// Staging areas of the packing threads, kept between calls: 128 threads that each map (and fault in) 2.5 MB per call
// serialise on the process's address-space lock -- the first version of the one-pass form was slower with 128 threads
// (0.6 - 1.7 s per 16 GB) than with 32 (0.23 - 0.35 s).
struct HostStagePool {
    std::mutex mu;
    std::vector<std::pair<unsigned char*, size_t>> idle;
    size_t idle_bytes = 0;
    unsigned char* take(size_t bytes, size_t* got) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].second >= bytes) {
                    unsigned char* p = idle[i].first;
                    *got = idle[i].second;
                    idle_bytes -= idle[i].second;
                    idle[i] = idle.back();
                    idle.pop_back();
                    return p;
                }
        }
        *got = bytes;
        return static_cast<unsigned char*>(std::malloc(bytes));
    }
    void give(unsigned char* p, size_t bytes) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (idle_bytes + bytes <= (size_t)1 << 30 && idle.size() < 512) {
                idle.emplace_back(p, bytes);
                idle_bytes += bytes;
                return;
            }
        }
        std::free(p);
    }
};
HostStagePool g_stage_pool;

template <typename T, typename B>
int host_pack_fused(const T* x, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t* indptr, int32_t* indices, T* values,
                    int64_t capacity, int n_threads, int64_t* total_out) {
    const int64_t n_blocks = (n_rows + kFusedRows - 1) / kFusedRows;
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n_blocks) n_threads = (int)(n_blocks > 0 ? n_blocks : 1);
    std::atomic<int64_t> next_claim{0}, next_commit{0};
    int64_t total = 0;  // written by the ticket holder only
    std::atomic<bool> overflow{false}, failed{false};
    const bool vec = host_has_avx512();
    indptr[0] = 0;
    auto worker = [&]() {
        const size_t cap = (size_t)kFusedRows * (size_t)n_cols + 32;
        size_t got = 0;
        unsigned char* raw = g_stage_pool.take(cap * (sizeof(int32_t) + sizeof(T)) + 64, &got);
        if (!raw) {
            overflow.store(true);  // (reported as out of memory by the caller; the tickets below must still be passed on)
            failed.store(true);
        }
        struct Give {
            unsigned char* p;
            size_t n;
            ~Give() { g_stage_pool.give(p, n); }
        } give_back{raw, got};
        T* s_val_p = reinterpret_cast<T*>(raw);  // (values first: 8-byte aligned)
        int32_t* s_idx_p = reinterpret_cast<int32_t*>(raw + cap * sizeof(T));
        int64_t cnt[kFusedRows];
        for (;;) {
            const int64_t k = next_claim.fetch_add(1, std::memory_order_relaxed);
            if (k >= n_blocks) return;
            const int64_t r0 = k * kFusedRows, r1 = r0 + kFusedRows < n_rows ? r0 + kFusedRows : n_rows;
            int64_t n = 0;
            for (int64_t r = r0; r < r1 && raw; ++r) {
                const T* xr = x + r * ld;
                int64_t c;
#if defined(__x86_64__)
                if (vec) {
                    if constexpr (sizeof(T) == 4) c = host_pack_row_avx512_f32((const float*)xr, n_cols, s_idx_p + n, (float*)s_val_p + n);
                    else c = host_pack_row_avx512_f64((const double*)xr, n_cols, s_idx_p + n, (double*)s_val_p + n);
                } else
#endif
                    c = host_pack_row_scalar<T, B>(xr, n_cols, s_idx_p + n, s_val_p + n);
                cnt[r - r0] = c;
                n += c;
            }
            if (!raw)
                for (int64_t r = r0; r < r1; ++r) cnt[r - r0] = 0;
            // the ticket: blocks take their place in block order (a few loads, then the thread steps aside -- 128
            // threads spinning on one cache line slow down the one that has to write it)
            for (int spins = 0; next_commit.load(std::memory_order_acquire) != k; ++spins) {
                if (spins < 64) {
#if defined(__x86_64__)
                    _mm_pause();
#endif
                } else {
                    std::this_thread::yield();
                }
            }
            const int64_t off = total;
            total = off + n;
            int64_t run = off;
            for (int64_t r = r0; r < r1; ++r) {
                run += cnt[r - r0];
                indptr[r + 1] = run;
            }
            const bool fits = off + n <= capacity;
            if (!fits) overflow.store(true, std::memory_order_relaxed);
            next_commit.store(k + 1, std::memory_order_release);
            if (fits && n) {
                std::memcpy(indices + off, s_idx_p, (size_t)n * sizeof(int32_t));
                std::memcpy(values + off, s_val_p, (size_t)n * sizeof(T));
            }
        }
    };
    if (n_threads == 1) {
        worker();
    } else {
        // (a thread that cannot be created -- ulimit, exhaustion -- must not leave joinable threads behind: the blocks are
        // claimed by whoever runs, so the threads that did start, and this one, finish the piece: ADVICE r5)
        std::vector<std::thread> pool;
        pool.reserve(n_threads);
        bool short_of_threads = false;
        for (int t = 0; t < n_threads && !short_of_threads; ++t) {
            try {
                pool.emplace_back(worker);
            } catch (const std::system_error&) {
                short_of_threads = true;
            }
        }
        if (short_of_threads) worker();
        for (auto& th : pool) th.join();
    }
    *total_out = total;
    if (failed.load()) throw std::bad_alloc();
    return overflow.load() ? 1 : 0;
}

template <typename F>
void host_parallel_rows(int64_t n_rows, int n_threads, F&& fn) {
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n_rows) n_threads = (int)(n_rows > 0 ? n_rows : 1);
    if (n_threads == 1) {
        fn((int64_t)0, n_rows);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(n_threads);
    for (int t = 0; t < n_threads; ++t) {
        const int64_t a = n_rows * t / n_threads, b = n_rows * (t + 1) / n_threads;
        try {
            pool.emplace_back([&fn, a, b] { fn(a, b); });
        } catch (const std::system_error&) {
            fn(a, b);  // no thread to be had: this range on the calling thread
        }
    }
    for (auto& th : pool) th.join();
}
}  // namespace

extern "C" {

int icv_host_dense_row_nnz(const void* h_x, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                           int64_t* h_row_nnz, int32_t n_threads) {
    if (!h_x || !h_row_nnz || n_rows < 0 || n_cols < 0 || ld < n_cols || (dtype != ICV_F32 && dtype != ICV_F64))
        return fail(ICV_ERR_INVALID, "bad host_dense_row_nnz arguments");
    try {
        const bool vec = host_has_avx512();
        if (dtype == ICV_F32)
            host_parallel_rows(n_rows, n_threads, [&](int64_t a, int64_t b) {
#if defined(__x86_64__)
                if (vec) return host_count_rows_avx512_f32((const float*)h_x, a, b, n_cols, ld, h_row_nnz);
#endif
                host_count_rows<float, uint32_t>((const float*)h_x, a, b, n_cols, ld, h_row_nnz);
            });
        else
            host_parallel_rows(n_rows, n_threads, [&](int64_t a, int64_t b) {
#if defined(__x86_64__)
                if (vec) return host_count_rows_avx512_f64((const double*)h_x, a, b, n_cols, ld, h_row_nnz);
#endif
                host_count_rows<double, uint64_t>((const double*)h_x, a, b, n_cols, ld, h_row_nnz);
            });
    } catch (const std::exception& e) {
        return fail(ICV_ERR_NOMEM, std::string("host_dense_row_nnz: ") + e.what());
    }
    return ICV_OK;
}

int icv_host_dense_pack(const void* h_x, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                        const int64_t* h_indptr, int32_t* h_indices, void* h_values, int32_t n_threads) {
    if (!h_x || !h_indptr || !h_indices || !h_values || n_rows < 0 || n_cols < 0 || ld < n_cols ||
        (dtype != ICV_F32 && dtype != ICV_F64))
        return fail(ICV_ERR_INVALID, "bad host_dense_pack arguments");
    try {
        const bool vec = host_has_avx512();
        if (dtype == ICV_F32)
            host_parallel_rows(n_rows, n_threads, [&](int64_t a, int64_t b) {
#if defined(__x86_64__)
                if (vec)
                    return host_pack_rows_avx512_f32((const float*)h_x, a, b, n_cols, ld, h_indptr, h_indices,
                                                     (float*)h_values);
#endif
                host_pack_rows<float, uint32_t>((const float*)h_x, a, b, n_cols, ld, h_indptr, h_indices, (float*)h_values);
            });
        else
            host_parallel_rows(n_rows, n_threads, [&](int64_t a, int64_t b) {
#if defined(__x86_64__)
                if (vec)
                    return host_pack_rows_avx512_f64((const double*)h_x, a, b, n_cols, ld, h_indptr, h_indices,
                                                     (double*)h_values);
#endif
                host_pack_rows<double, uint64_t>((const double*)h_x, a, b, n_cols, ld, h_indptr, h_indices,
                                                 (double*)h_values);
            });
    } catch (const std::exception& e) {
        return fail(ICV_ERR_NOMEM, std::string("host_dense_pack: ") + e.what());
    }
    return ICV_OK;
}

int icv_host_dense_pack_fused(const void* h_x, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                              int64_t* h_indptr, int32_t* h_indices, void* h_values, int64_t capacity,
                              int32_t n_threads, int64_t* h_nnz) {
    if (!h_x || !h_indptr || !h_nnz || n_rows < 0 || n_cols < 0 || ld < n_cols || capacity < 0 ||
        (capacity > 0 && (!h_indices || !h_values)) || (dtype != ICV_F32 && dtype != ICV_F64))
        return fail(ICV_ERR_INVALID, "bad host_dense_pack_fused arguments");
    try {
        int over;
        if (dtype == ICV_F32)
            over = host_pack_fused<float, uint32_t>((const float*)h_x, n_rows, n_cols, ld, h_indptr, h_indices,
                                                    (float*)h_values, capacity, n_threads, h_nnz);
        else
            over = host_pack_fused<double, uint64_t>((const double*)h_x, n_rows, n_cols, ld, h_indptr, h_indices,
                                                     (double*)h_values, capacity, n_threads, h_nnz);
        if (over)
            return fail(ICV_ERR_NOMEM, "host_dense_pack_fused: the matrix holds more stored entries than `capacity` "
                                       "(*h_nnz has the count: call again with larger buffers)");
    } catch (const std::exception& e) {
        return fail(ICV_ERR_NOMEM, std::string("host_dense_pack_fused: ") + e.what());
    }
    return ICV_OK;
}

int icv_csr_scatter_dense(const void* data, int32_t dtype, const int64_t* indptr, const int32_t* indices, int64_t n_rows,
                          int32_t n_cols, void* out, int64_t ldo, void* stream) {
    if (!indptr || !out || n_rows < 0 || n_cols < 0 || ldo < n_cols || (dtype != ICV_F32 && dtype != ICV_F64))
        return fail(ICV_ERR_INVALID, "bad csr_scatter_dense arguments");
    if (n_rows == 0 || n_cols == 0) return ICV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t esz = dtype == ICV_F32 ? 4 : 8;
    if (ldo == n_cols)
        HIP_TRY(hipMemsetAsync(out, 0, (size_t)n_rows * n_cols * esz, st));
    else
        HIP_TRY(hipMemset2DAsync(out, (size_t)ldo * esz, 0, (size_t)n_cols * esz, (size_t)n_rows, st));
    dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
    if (dtype == ICV_F32)
        hipLaunchKernelGGL(icv::k_csr_scatter_rows<float>, grid, block, 0, st, static_cast<const float*>(data), indptr,
                           indices, n_rows, static_cast<float*>(out), ldo);
    else
        hipLaunchKernelGGL(icv::k_csr_scatter_rows<double>, grid, block, 0, st, static_cast<const double*>(data), indptr,
                           indices, n_rows, static_cast<double*>(out), ldo);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

int icv_group_sums(const double* values, const int32_t* group, int64_t n, int32_t n_groups, double* sums,
                   int64_t* counts, void* stream) {
    if (!sums || !counts || n < 0 || n_groups < 0 || (n > 0 && (!values || !group)))
        return fail(ICV_ERR_INVALID, "bad group_sums arguments");
    if (n_groups == 0) return ICV_OK;
    hipLaunchKernelGGL(icv::k_group_sums, dim3((unsigned)n_groups), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       values, group, n, sums, counts);
    HIP_TRY(hipGetLastError());
    return ICV_OK;
}

}  // extern "C"

/*
 * infercnv_hip.h -- C ABI of the MI355X (gfx950) CNV-inference engine.
 *
 * This is the drop-in boundary for ONE hot path of icbi-lab/infercnvpy:
 * the per-chunk numeric kernel `_infercnv_chunk` (reference
 * src/infercnvpy/tl/_infercnv.py:411-457) together with the pieces of the
 * `infercnv` driver that touch every matrix element (reference mean,
 * :359-408; chunk fan-out/vstack, :120-139) and the `cnv_score` reduction
 * (src/infercnvpy/tl/_scores.py:65-68).
 *
 * The reference has no FFI; its seam is a Python function called once per
 * 5000-cell chunk from a ProcessPoolExecutor worker.  A maintainer binds this
 * library with ctypes and calls it from `infercnv()` in
 * place of `process_map(_infercnv_chunk, ...)`.
 *
 * Conventions
 *   - plain C types only; every data pointer is a DEVICE pointer (HBM) unless
 *     the parameter name starts with `h_` (host);
 *   - every function returns an icv_status (0 = ok); `icv_last_error()` returns
 *     a thread-local message for the last failure; no C++ exception crosses;
 *   - the caller owns all buffers; the only library-owned object is the opaque
 *     plan (create / destroy);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     are asynchronous on that stream unless stated otherwise.
 */
#ifndef INFERCNV_HIP_H
#define INFERCNV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ICV_OK = 0,
    ICV_ERR_INVALID = 1,     /* bad argument (maps to Python ValueError)            */
    ICV_ERR_UNSUPPORTED = 2, /* configuration exceeds an implementation limit       */
    ICV_ERR_HIP = 3,         /* HIP runtime failure (message has the hipError name) */
    ICV_ERR_NOMEM = 4
} icv_status;

enum { ICV_F32 = 0, ICV_F64 = 1 };
enum { ICV_DENSE = 0, ICV_CSR = 1 };

/* flags for icv_infercnv_* */
enum {
    ICV_FLAG_TRUNC_TO_INT = 1, /* bounded centring of an integer matrix keeps the matrix dtype
                                  (reference :428): centred values are truncated toward zero */
    ICV_FLAG_ROUND_F32 = 2,    /* bounded centring of a float32 matrix against a float64 reference:
                                  differences are rounded to float32 (same line)              */
    ICV_FLAG_NO_APPLY = 4      /* icv_infercnv_run: compute the thresholds, leave `out` un-thresholded */
};

/* A cells x genes expression matrix resident in HBM. */
typedef struct {
    int32_t format;         /* ICV_DENSE | ICV_CSR                                     */
    int32_t dtype;          /* ICV_F32 | ICV_F64 (dtype of `values`)                   */
    int64_t n_rows;         /* cells                                                   */
    int32_t n_cols;         /* genes = n_cols_all of the plan                          */
    int32_t _pad;           /* dense: column offset of `values` inside its row when the matrix is a column view of a
                               wider one (0 otherwise); only icv_colchain looks at it (which loads stay in bounds).
                               csr: optional hint, the number of stored entries MOST rows stay under (e.g. the 99.9 %
                               quantile of the row lengths; 0 = unknown): the stored-entries kernel sizes its per-cell
                               entry slots with it -- a performance hint only, any value gives the same results      */
    int64_t ld;             /* dense: row stride in elements (>= _pad + n_cols)        */
    const void *values;     /* dense: n_rows x ld row-major; csr: nnz values           */
    const int64_t *indptr;  /* csr: n_rows + 1                                         */
    const int32_t *indices; /* csr: nnz column indices, unique and ascending within a row */
    int64_t csr_begin;      /* csr: host copies of indptr[0] and indptr[n_rows]; both 0 = unknown  */
    int64_t csr_end;        /*      (the prepared-entry fast path needs them to size its workspace) */
} icv_matrix;

typedef struct icv_plan_s *icv_plan_t;

typedef struct {
    int32_t n_cols_all;   /* columns of the input matrix                               */
    int32_t n_genes_used; /* columns that fall in a window-bearing chromosome          */
    int32_t n_chr;
    int32_t window;
    int32_t step;
    int32_t n_windows;    /* W = sum_c W_c  (reference :218, :227-236)                  */
    int32_t block;        /* B: genes per partial-sum block (1 = direct form)          */
    int32_t n_blocks;     /* padded blocks over all chromosomes                        */
    int32_t padded_len;   /* LDS row length in elements                                */
    int32_t lds_bytes_f32;/* dynamic LDS per workgroup for float32 input               */
    int32_t lds_bytes_f64;
    int32_t workgroups_per_cu_f32;
} icv_plan_info;

/* ---- planning (host only, no GPU needed) -------------------------------------------------
 * Gene/position indexing contract (reference :327, :350-351): the caller sorts the genes of
 * every window-bearing chromosome by `start` and concatenates chromosomes in natural order.
 *   h_col_pos[g]      position of input column g in that concatenation, or -1 if the column is
 *                     masked (no position / excluded chromosome / chrM / non-"chr" contig)
 *   h_chrom_offsets   n_chr + 1 offsets into the concatenation
 * The library derives the window table: per chromosome with G_c genes,
 *   window < G_c : W_c = ceil((G_c - window + 1) / step) pyramid windows  (:205-218)
 *   otherwise    : one flat window over all G_c genes                     (:227-236)
 * Concurrency: the tables of a plan are immutable after creation, but the plan also owns the per-call device
 * workspace of the compute entry points (moment partials, hand-back lists, CSR staging): ONE compute call at a
 * time per plan.  A second call entering while one is in its launch sequence fails with ICV_ERR_INVALID ("plan
 * busy") instead of racing.  Successive calls on one plan are ordered on the device: a call enqueued on another
 * stream than the plan's previous call first waits (hipStreamWaitEvent) for that call's kernels, so the shared
 * workspace is never used by two calls at once; calls on different plans are independent.
 */
int icv_plan_create(int32_t n_cols_all, const int32_t *h_col_pos, int32_t n_chr,
                    const int32_t *h_chrom_offsets, int32_t window, int32_t step, icv_plan_t *out);
void icv_plan_destroy(icv_plan_t plan);
int icv_plan_get_info(icv_plan_t plan, icv_plan_info *h_info);
/* first window index of every chromosome = the values of uns["cnv"]["chr_pos"] (:335-337) */
int icv_plan_chr_pos(icv_plan_t plan, int32_t *h_chr_pos /* n_chr */);
/* window table in sorted-gene coordinates: window j covers [start, start+len) */
int icv_plan_window_table(icv_plan_t plan, int32_t *h_start /* W */, int32_t *h_len /* W */);
/* Which smoothing kernel the plan's last compute call (icv_infercnv_smooth / _run / icv_gene_values) launched:
 * diagnostics and tests ("this input really took the CSR stored-entries kernel"). */
enum {
    ICV_KERNEL_NONE = 0,
    ICV_KERNEL_GENERIC = 1, /* k_smooth: any dtype / format / geometry that fits LDS                      */
    ICV_KERNEL_WS = 2,      /* k_smooth_ws: dense float32, block form                                     */
    ICV_KERNEL_WS_CSR = 3,  /* k_csr_prepare + k_smooth_ws: CSR float32, row rebuilt in LDS               */
    ICV_KERNEL_X16 = 4,     /* k_smooth_x16: dense float32, window 100 or 250 / step 10                   */
    ICV_KERNEL_SD = 5,      /* k_smooth_sd: CSR float32, block form, stored entries only (prefix sums)    */
    ICV_KERNEL_SPLIT = 6    /* chromosome groups (row larger than LDS) + median on float64 windows in HBM */
};
int icv_plan_last_kernel(icv_plan_t plan, int32_t *h_kind);
/* Host tables of the CSR stored-entries kernel (k_smooth_se), for tests that restate its arithmetic on the CPU:
 * *h_applies = 1 if the plan's geometry admits the kernel (else nothing more is written); per input column the
 * block of its gene (-1: masked) and the gene's offset inside the block; per block the offset of its first gene
 * inside its chromosome; per window the two packed words that name the LDS slots of its three prefix sums and the
 * wavefront totals to add (csrc/icv_plan.hpp: se_window_words).  Any of the array pointers may be NULL. */
int icv_plan_se_tables(icv_plan_t plan, int32_t *h_applies, int32_t *h_col_block /* n_cols_all */,
                       int32_t *h_col_offset /* n_cols_all */, int32_t *h_block_gene0 /* n_blocks */,
                       uint32_t *h_w0 /* W */, uint32_t *h_w1 /* W */);
/* Host tables of calculate_gene_values (reference tl/_infercnv.py:247-298: a gene's value is the mean of the kept windows
 * that contain it), for tests that check them on the CPU against the oracle: the covered genes in chromosome / position
 * order fall into RUNS that share their windows; run r averages the windows [first, first + count) (global window
 * indices) and holds `genes` genes; h_col_run[c] = the run of input column c, -1 where the gene has no value (masked
 * chromosome, or no kept window contains it: NaN in the layer, :147).  *h_n_runs is always written; the arrays (each may
 * be NULL) hold *h_n_runs / n_cols_all entries: call once with NULL arrays for the size. */
int icv_plan_gene_runs(icv_plan_t plan, int32_t *h_n_runs, int32_t *h_run_first, int32_t *h_run_count,
                       int32_t *h_run_genes, int32_t *h_col_run /* n_cols_all */);

/* ---- reference profile (reference :385, :400) --------------------------------------------
 * Per-group column sums in float64.  h/d: `row_group` (device, n_rows int32; -1 = row not in
 * any group; NULL = every row in group 0).  `sums` (device, n_groups x n_cols float64) is
 * ACCUMULATED into, so shards / ranks can add up before the caller divides by the counts.
 * Dense and CSR sums are deterministic (fixed reduction order; CSR: a 1024-thread workgroup per row slab adds ONE
 * row at a time into float64 LDS accumulators, a barrier between rows, so every column receives its addends in
 * row order; the slabs are then added in slab order).
 */
int icv_colsum(const icv_matrix *m, const int32_t *row_group, int32_t n_groups, double *sums,
               void *stream);

/* The same profile in the REFERENCE'S OWN EVALUATION ORDER, so that `reference=None` / `reference_cat` reproduce the
 * reference bit for bit (float32 sums are not associative):
 *   - np.mean(X, axis=0) of a C-contiguous matrix (reference :385, :400) is, per column, the sequential chain
 *     acc = fl(acc + x[r][g]) over the rows in order, in the matrix dtype (float32 stays float32), then acc / n;
 *   - scipy's CSR mean is the same chain over fl(x * fl(1/n)) (no division afterwards);
 *   - scipy's CSC mean is np.add.reduceat per column: first stored entry + numpy's pairwise sum of the others;
 *   - np.mean(X, axis=0) of a dense matrix stored column-major: pairwise per column (icv_colsum_pairwise below).
 * icv_colchain continues the chains: `acc` (device, n_cols values of the MATRIX dtype; zero before the first call) is
 * read, the rows `rows[0..n_sel)` of `m` (device int32, ascending; NULL = all rows of m) are added in order, and the
 * new accumulators are written back -- row pieces, slabs and shards are chained by calling in row order with the same
 * `acc`.  `scale` is used for CSR input only (1 / n of the whole group, rounded to the matrix dtype inside).
 * Columns are the only parallelism an exact chain has: one workgroup per 128..512-byte tile of a row streams its rows
 * through an LDS ring (LDS-DMA) while one wavefront adds them in order.  CSR needs 4 * (tiles + 1) bytes of temporary
 * device memory per row (stream-ordered).  Nothing is read back: the call never synchronises.
 * icv_colchain_mean: mean[c] = acc[c] / count in the dtype (dense); for CSR the accumulators already are the means.
 * icv_colmean_csc: means of a CSC matrix (colptr n_cols + 1 int64, row_idx int32, values), entries of rows with
 * row_group[row] == group only (row_group NULL: all), scale = 1 / n as above; `mean` n_cols values of the dtype. */
int icv_colchain(const icv_matrix *m, const int32_t *rows, int64_t n_sel, double scale, void *acc, void *stream);
/* The all-cell sums of a dense matrix stored COLUMN-major (numpy puts the axis with the smaller stride innermost:
 * np.mean(X, axis=0) of an F-ordered array reduces every column with its contiguous inner loop = pairwise summation,
 * over pieces of 8 192 elements, the iterator's buffer): `xt` holds n_cols columns of n_rows contiguous values each,
 * `ld` elements apart (the caller uploads column blocks as they lie in host memory); sums[c] = that reduction, in the
 * dtype (icv_colchain_mean divides). */
int icv_colsum_pairwise(const void *xt, int32_t dtype, int64_t n_rows, int32_t n_cols, int64_t ld, void *sums,
                        void *stream);
int icv_colchain_mean(const void *acc, int32_t dtype, int32_t n_cols, int64_t count, void *mean, void *stream);
int icv_colmean_csc(const void *values, int32_t dtype, const int64_t *colptr, const int32_t *row_idx, int32_t n_cols,
                    const int32_t *row_group, int32_t group, double scale, void *mean, void *stream);

/* ---- the hot path ------------------------------------------------------------------------
 * Steps 1-4 of `_infercnv_chunk` (:422-442) for every row of `m`, fused in one kernel:
 *   centre on the reference (ref_hi == NULL: x - ref_lo; else bounded difference),
 *   clip to +-lfc_clip in the matrix dtype, pyramid/flat windowed mean per chromosome in
 *   float64, subtract the per-cell median over all W windows.
 * Outputs (device):
 *   out         n_rows x ldo float32, un-thresholded x_res (rows on 16-byte boundaries -- ldo % 4 == 0 and a
 *               16-byte aligned base -- let the dense fast kernel store 16 bytes per lane)
 *   cell_median n_rows float64
 *   cell_stats  n_rows x 2 float64: sum(x_res), sum(x_res^2) of the row (for the chunk std)
 * `ref_lo`/`ref_hi`: n_cols values of the matrix dtype, in input column order.
 */
int icv_infercnv_smooth(icv_plan_t plan, const icv_matrix *m, const void *ref_lo, const void *ref_hi,
                        double lfc_clip, int32_t flags, float *out, int64_t ldo, double *cell_median,
                        double *cell_stats, void *stream);

/* Step 5a (:450): thr[k] = dynamic_threshold * population-std over chunk k, where chunk k is rows
 * [k*chunksize - row_phase, ...) of this shard (row_phase = global index of the shard's first row
 * modulo chunksize; 0 when shards are chunk-aligned).  `thr` device float64[n_chunks]. */
int icv_chunk_thresholds(const double *cell_stats, int64_t n_rows, int64_t chunksize, int64_t row_phase,
                         int32_t n_windows, double dynamic_threshold, double *thr, void *stream);

/* Step 5b (:451): out[|x| < thr[chunk(row)]] = 0, decided in float64: an entry whose float32
 * value ties with float32(thr) is recomputed from the input in float64 before deciding. */
int icv_apply_threshold(icv_plan_t plan, const icv_matrix *m, const void *ref_lo, const void *ref_hi,
                        double lfc_clip, int32_t flags, float *out, int64_t ldo,
                        const double *cell_median, const double *thr, int64_t chunksize,
                        int64_t row_phase, void *stream);

/* Timings of the last icv_infercnv_run with profiling enabled (milliseconds, HIP events
 * recorded on `stream`). */
typedef struct {
    float smooth_ms;
    float thresholds_ms;
    float apply_ms;
    float total_ms;
} icv_profile;

/* Convenience: smooth -> chunk thresholds -> apply, on one stream.  `dynamic_threshold` NaN
 * disables step 5 (reference: None).  `thr` may be NULL only when step 5 is disabled.
 * `cell_stats` may be NULL: the per-cell moments are then not returned, and where the dense fast kernel runs
 * they are not formed at all (the kernel accumulates the noise-threshold moments per chunk).
 * flags | ICV_FLAG_NO_APPLY: compute `thr` but leave `out` un-thresholded (for icv_threshold_mask).
 * One call at a time per plan: the plan owns the per-call workspace (see icv_plan_create).
 * If `h_profile` is non-NULL the call records HIP events around each stage, synchronises the
 * stream before returning and fills the struct. */
int icv_infercnv_run(icv_plan_t plan, const icv_matrix *m, const void *ref_lo, const void *ref_hi,
                     double lfc_clip, double dynamic_threshold, int64_t chunksize, int64_t row_phase,
                     int32_t flags, float *out, int64_t ldo, double *cell_median, double *cell_stats,
                     double *thr, icv_profile *h_profile, void *stream);

/* Deferred timing for back-to-back runs: after icv_profile_begin every icv_infercnv_run on this plan (called
 * with h_profile == NULL) records HIP events around its stages on its stream WITHOUT synchronising;
 * icv_profile_collect waits for the recorded runs, fills up to max_records structs in call order, returns the
 * count and ends the deferred mode.  (bench.py: kernel time measured over the timed region with no host
 * synchronisation between steps.) */
int icv_profile_begin(icv_plan_t plan);
int icv_profile_collect(icv_plan_t plan, icv_profile *out, int32_t max_records, int32_t *n_records);

/* ---- the reference-order float32 chain by BLOCKS of rows (csrc/icv_kernel_blocks.hpp) ---------------------------
 * icv_colchain evaluates np.mean's sequential float32 chain (reference :385) at HBM stream rate on ONE GPU, but row
 * shards on several GPUs must take turns.  Inside one binade the chain is integer arithmetic (fl(s + x) = s + q ulp(s)
 * with q = round(x / ulp(s)) independent of s except at exact ties), so a block of 64 rows whose chain stays inside its
 * binade and holds no tie is ONE integer: the ranks compute them CONCURRENTLY from a float64 estimate of their start,
 * and only a scan over the records (plus the rare replays) runs in row order.  Dense float32 matrices, all rows;
 * everything else: ICV_ERR_UNSUPPORTED (use icv_colchain).  Bit-equal to icv_colchain for any estimate.
 *   workspace: device buffer of icv_colchain_blocks_workspace(n_rows, n_cols) bytes, the same for the three calls
 *   1. icv_colchain_blocks_sums:    float64 column totals of the matrix -> total[n_cols] (device); all-gather them
 *   2. icv_colchain_blocks_records: est_start[n_cols] (device float64, NULL = 0) = the estimate of the chain value
 *                                   before row 0 (sum of the earlier ranks' totals) -> the block records
 *   3. icv_colchain_blocks_scan:    acc[n_cols] (device float32, in / out) = the EXACT chain values before row 0 (from
 *                                   the previous rank) -> after the last row; columns [col0, col1) only (ranks pipeline
 *                                   over column groups); d_replayed (device, optional) += blocks replayed row by row. */
int icv_colchain_blocks_workspace(int64_t n_rows, int32_t n_cols, int64_t *bytes);
int icv_colchain_blocks_sums(const icv_matrix *m, void *workspace, double *total, void *stream);
int icv_colchain_blocks_records(const icv_matrix *m, void *workspace, const double *est_start, void *stream);
int icv_colchain_blocks_scan(const icv_matrix *m, void *workspace, float *acc, int32_t col0, int32_t col1,
                             uint64_t *d_replayed, void *stream);

/* ---- calculate_gene_values=True (reference :247-298, :443-453) -------------------------------
 * Per-gene CNV values: mean of the kept windows that contain the gene (:274-288), minus the per-cell median
 * over the covered genes (:443-444), zeroed below the chunk's noise threshold (:452-453; `thr` from
 * icv_chunk_thresholds / icv_infercnv_run; NULL = no thresholding).  `gene_out` (device, float64,
 * n_rows x ldg, ldg >= n_cols) is written completely: NaN for genes no kept window covers and for
 * masked genes (reference: reindex with NaN fill, :147).
 *
 * The windows a gene averages are the float64 windows of step 3 BEFORE the per-cell centring -- what the
 * smoothing kernel holds anyway.  icv_infercnv_run_windows is icv_infercnv_run that also writes them
 * (`win_out`: device float64, n_rows x ldw, ldw >= n_windows; NULL = plain icv_infercnv_run): the same kernel
 * as the plain call (k_smooth_x16 / k_smooth_se store them beside x_res; other geometries: the generic kernel),
 * 8 * n_windows more bytes per cell.  icv_gene_values_from_windows turns them into the gene layer in ONE kernel:
 * windows -> LDS, one value per run of genes that share their windows, weighted median by radix select, the
 * row written once in input-column order.  So `calculate_gene_values=True` costs the plain call + the windows +
 * one pass that writes 8 * n_cols bytes per cell -- no second smoothing (reference: _infercnv_chunk returns both
 * from one pass too, :438-457).
 *
 * icv_gene_values (kept): smoothing + gene layer in one call for callers that do not need X_cnv;
 * stream-ordered temporaries of 12 * n_windows bytes per row. */
int icv_infercnv_run_windows(icv_plan_t plan, const icv_matrix *m, const void *ref_lo, const void *ref_hi,
                             double lfc_clip, double dynamic_threshold, int64_t chunksize, int64_t row_phase,
                             int32_t flags, float *out, int64_t ldo, double *cell_median, double *cell_stats,
                             double *thr, icv_profile *h_profile, double *win_out, int64_t ldw, void *stream);
int icv_gene_values_from_windows(icv_plan_t plan, const double *win, int64_t ldw, int64_t n_rows,
                                 const double *thr, int64_t chunksize, int64_t row_phase, double *gene_out,
                                 int64_t ldg, void *stream);
int icv_gene_values(icv_plan_t plan, const icv_matrix *m, const void *ref_lo, const void *ref_hi,
                    double lfc_clip, int32_t flags, const double *thr, int64_t chunksize,
                    int64_t row_phase, double *gene_out, int64_t ldg, void *stream);

/* ---- CSR packing of X_cnv (reference :455 `csr_matrix(x_res)`, :137 vstack) ------------------
 * Two steps around a caller-side prefix sum: count the non-zeros of every row of the dense float32
 * result, then write column indices (int32) and values (float64) at indptr[row]. */
int icv_csr_count(const float *x, int64_t n_rows, int32_t n_cols, int64_t ld, int64_t *row_nnz, void *stream);
int icv_csr_fill(const float *x, int64_t n_rows, int32_t n_cols, int64_t ld, const int64_t *indptr,
                 int32_t *indices, double *data, void *stream);

/* Geometry of the streamed pack kernels for rows of n_windows float32 (host logic, no GPU needed): whether
 * icv_threshold_mask / icv_csr_fill_masked take the LDS-ring kernels at this width (given aligned rows), how the 160 KB
 * of LDS of a CU are split, and how many LDS-DMA loads a loader wavefront keeps in flight (the hardware counter holds 63). */
typedef struct {
    int32_t rows_per_round;        /* adjacent rows a workgroup takes per barrier */
    int32_t mask_streamed;         /* 1: k_thr_mask_ring applies */
    int32_t mask_ring_slots;       /* ring slots of rows_per_round rows (one being read, the others in flight) */
    int32_t mask_lds_bytes;        /* ring + the two staging blocks of mask words and counts */
    int32_t mask_loads_in_flight;  /* per loader wavefront */
    int32_t fill_streamed;         /* 1: k_csr_fill_ring applies */
    int32_t fill_ring_slots;
    int32_t fill_lds_bytes;        /* ring (rows + the round's mask rows and row offsets) + two staging blocks */
    int32_t fill_loads_in_flight;  /* loader 0 (it also brings the mask rows and row offsets) */
    int32_t fill_stage_entries;    /* kept windows per staging block: a round with more takes passes */
} icv_pack_info;
int icv_pack_geometry(int32_t n_windows, icv_pack_info *h_info);

/* Step 5b fused with the packing (the public tl.infercnv path: X_cnv leaves the GPU as CSR and the dense thresholded
 * matrix is never needed): same decision as icv_apply_threshold, but `out` (the UN-thresholded x_res of
 * icv_infercnv_smooth / icv_infercnv_run with ICV_FLAG_NO_APPLY) is left untouched; the kept entries
 * (|x| >= thr in float64, x != 0; NaN kept) are recorded as bits, `mask` = n_rows x n_words uint64 with
 * n_words = ceil(n_windows / 64), and counted into `row_nnz`.  `thr` NULL = no threshold (dynamic_threshold=None).
 * icv_row_offsets turns row_nnz into indptr, icv_csr_fill_masked then packs indices / values at indptr[row]: the
 * public path's three calls, nothing is read back.  x_res is read twice in total and never rewritten.
 * Both passes stream x_res through an LDS ring (one persistent workgroup per CU, LDS-DMA loader wavefronts:
 * k_thr_mask_ring / k_csr_fill_ring, csrc/icv_kernel_pack.hpp) when its rows are 16-byte aligned (ldo a multiple of 4)
 * and hold 256 .. 2 048 windows (four rows a round through 160 KB of LDS); other shapes run one workgroup / wavefront
 * per row.  The streamed mask pass needs 24 * n_rows bytes of temporary device memory (stream-ordered) when `thr` is
 * given. */
int icv_threshold_mask(icv_plan_t plan, const icv_matrix *m, const void *ref_lo, const void *ref_hi,
                       double lfc_clip, int32_t flags, const float *out, int64_t ldo, const double *cell_median,
                       const double *thr, int64_t chunksize, int64_t row_phase, uint64_t *mask,
                       int64_t *row_nnz, void *stream);
int icv_csr_fill_masked(const float *x, int64_t n_rows, int32_t n_cols, int64_t ld, const uint64_t *mask,
                        const int64_t *indptr, int32_t *indices, double *data, void *stream);
/* the prefix sum between the two: indptr[0] = 0, indptr[r + 1] = row_nnz[0] + ... + row_nnz[r] (device arrays; two
 * launches over blocks of 4 096 rows, 8 bytes of temporary device memory per block) */
int icv_row_offsets(const int64_t *row_nnz, int64_t n_rows, int64_t *indptr, void *stream);

/* Step 5b and the packing in ONE pass (kept for comparison: 0.7 ms per 100 000 cells of config 2 against 0.33 ms for
 * the three calls above, profiles/r04_pack_experiments.txt): the decision of
 * icv_threshold_mask, the rows' offsets from a two-level decoupled look-back over the rows (no mask array, no prefix-sum call,
 * no second kernel) and the kept entries written from the row: x_res is read from HBM once.  `indptr` n_rows + 1
 * offsets starting at 0; `indices` / `data` hold `capacity` entries (n_rows * n_windows can never overflow; entries
 * past the capacity are dropped, indptr[n_rows] still tells how many there are).  Deterministic: the output does not
 * depend on the order in which the rows finish.  At most 20 480 windows (ICV_ERR_UNSUPPORTED beyond: use the
 * two-step form).  Needs less than n_rows bytes of temporary device memory (stream-ordered). */
int icv_threshold_pack(icv_plan_t plan, const icv_matrix *m, const void *ref_lo, const void *ref_hi, double lfc_clip,
                       int32_t flags, const float *out, int64_t ldo, const double *cell_median, const double *thr,
                       int64_t chunksize, int64_t row_phase, int64_t *indptr, int32_t *indices, double *data,
                       int64_t capacity, void *stream);

/* ---- ithcna / ithgex (tl/_scores.py:77-221) -------------------------------------------------
 * Interquartile range of all n x n entries of np.corrcoef(x) for one group of cells: x is a dense
 * float32 n x k matrix in HBM (rows = cells).  Rows are centred / scaled with float64 statistics, the
 * Gram matrix is computed with fp32 MFMA tiles, the percentiles (numpy "linear" method) from exactly
 * selected order statistics.  Synchronous (the selection reads counters back); *h_iqr is a host double.
 * Needs 4*n*(n + k) bytes of temporary device memory. */
int icv_corr_iqr(const float *x, int64_t n, int32_t k, int64_t ld, double *h_iqr, void *stream);

/* ---- cell-level hierarchical clustering behind the heatmap dendrogram (BASELINE.json config 5) ----
 * No call site in the reference (scanpy's `dendrogram=True` of sc.pl.heatmap, forwarded by
 * pl/_chromosome_heatmap.py:74-85, clusters category means); the oracle is scipy pdist + linkage("ward").
 *
 * icv_pairwise_sqeuclidean: out[(i - row_begin)*ldo + j] = ||x_i - x_j||^2 for row_begin <= i < row_end and
 * all 0 <= j < n (float32; the full symmetric n x n matrix with a zero diagonal for row range [0, n); a row
 * block is what one GPU of a row-sharded job computes).  x is a dense float32 n x d matrix in HBM; columns are
 * centred first (translation invariant), the contraction runs on fp32 MFMA tiles.  Needs 4*n*d + 8*n bytes
 * of temporary device memory.
 *
 * icv_ward_linkage: Ward linkage of the n points whose squared distances are in dist_sq (n x n float32 in
 * HBM, symmetric, row stride ld >= n; OVERWRITTEN).  spare_columns != 0: the caller also hands over columns
 * [n, ld) of every row as scratch (see "Column layout" below; ld % 4 == 0 and ld >= n + (n + 1) / 2 required, else
 * ICV_ERR_INVALID); 0: nothing outside the n x n block is touched, whatever the stride (dist_sq may be a column
 * slice of a wider buffer).  h_linkage is a HOST array of (n-1) x 4 doubles in scipy's linkage-matrix
 * format (cluster ids, height, size; rows sorted by height).  Synchronous; *h_rounds (optional) returns
 * the number of reciprocal-nearest-neighbour rounds.  ICV_ERR_INVALID if a distance is NaN. */
int icv_pairwise_sqeuclidean(const float *x, int64_t n, int32_t d, int64_t ld, int64_t row_begin, int64_t row_end,
                             float *out, int64_t ldo, void *stream);
int icv_ward_linkage(float *dist_sq, int64_t n, int64_t ld, int32_t spare_columns, double *h_linkage,
                     int32_t *h_rounds, void *stream);

/* The same two steps for a distance matrix SHARDED by rows over several GPUs (one process per GPU; the exchanges
 * between the calls are the caller's: infercnvpy_amd/dist.py does them with torch.distributed over RCCL).
 * Rows are owned in super-rows of ICV_SUPER_ROWS rows.
 *
 * icv_pairwise_sqeuclidean_tiles: this rank's share of the upper-triangular super-tiles (ICV_SUPER_ROWS x
 * ICV_SUPER_ROWS outputs each; x = ALL n points, resident).  Tile k covers rows row0[k].. and columns col0[k]..
 * (multiples of ICV_SUPER_ROWS, col0 >= row0); it is written to dir + dir_off[k] (row stride ld_dir) and, for the
 * part strictly above the diagonal, transposed to mir + mir_off[k] (row stride ld_mir): the rank writes its own
 * rows directly and the mirror blocks into the buffer that travels to the owners of those rows.  All arrays of
 * length n_tiles are HOST arrays.  Values are bit-identical to icv_pairwise_sqeuclidean's. */
#define ICV_SUPER_ROWS 1024
int icv_pairwise_sqeuclidean_tiles(const float *x, int64_t n, int32_t d, int64_t ld, int32_t n_tiles,
                                   const int32_t *h_row0, const int32_t *h_col0, const int64_t *h_dir_off,
                                   const int64_t *h_mir_off, float *dir, int64_t ld_dir, float *mir, int64_t ld_mir,
                                   void *stream);

/* Column layout of the Ward rounds (both entry points; chosen by the caller's spare_columns argument, never
 * inferred from the stride).  With spare columns -- a row stride that leaves at least n / 2 of them
 * (ld >= n + (n + 1) / 2, ld % 4 == 0) -- a merged cluster keeps its row but moves to a NEW column: the clusters
 * merged in a round take consecutive columns of the spare region, so that the update of every other row is one
 * contiguous strip instead of one 4-byte write per 128-byte line; the alive columns are compacted in place when
 * the region is full (such a matrix must be 16-byte aligned).  Without spare columns the columns are updated in
 * place.  Results are identical either way, bit for bit.
 *
 * Step-wise Ward rounds on a row-sharded matrix.  Every rank creates the same state (the bookkeeping is replicated
 * and deterministic); h_sr_local[g] = local super-row index of global super-row g on THIS rank (0 .. k-1) or -1
 * (NULL: all rows are local); ld = row stride of the local matrix, the same in every later call.  One round:
 *   icv_ward_merge     rows merged in the previous round, by their owners: new row, its nearest neighbour;
 *                      h_pslot[p] = row of `stage` holding the partner row of merge p (received from its owner) or
 *                      -1 if the partner is stored locally; scatter != 0: also update the other local rows
 *                      (single-GPU form; sharded callers use gather / exchange / scatter instead)
 *   icv_ward_gather    out[q][k] = entry of local row d_rows[q] (a new row) for the cluster of slot d_slots[k]:
 *                      the columns another rank needs, in the order of ITS local rows (device index arrays)
 *   icv_ward_scatter   the update of the local rows that did not merge from the n_pairs new rows v[q][local row]
 *                      (row stride ldv); v[q] belongs to merge h_vrow_p[q] of the round (a permutation)
 *   icv_ward_scan      nearest neighbour of the local rows whose cached neighbour merged or died
 *   icv_ward_pack_nn / icv_ward_unpack_nn   the round's (neighbour, distance) results, n_pairs + n_act entries in
 *                      list order, zero where another rank owns the row: all-reduce(SUM) them in between
 *   icv_ward_pairs     reciprocal pairs of the round (replicated); compacts the local rows first when the column
 *                      layout asks for it; h_counts = {n_live, n_merges, n_pairs, n_act} (synchronises the
 *                      stream); all_active != 0: list every live row for the next scan
 *   icv_ward_round_pairs  slots (i kept, j absorbed) of the pairs just found, HOST arrays of n_pairs
 * until n_live == 1; icv_ward_finish writes the linkage matrix as icv_ward_linkage does.  One in-flight call per
 * state. */
typedef struct icv_ward_s *icv_ward_t;
int icv_ward_create(int64_t n, const int32_t *h_sr_local, int32_t n_super, int32_t super_shift, int64_t ld,
                    int32_t spare_columns, icv_ward_t *out, void *stream);
void icv_ward_destroy(icv_ward_t w);
int icv_ward_merge(icv_ward_t w, float *d_local, int64_t ld, const float *stage, int64_t ld_stage,
                   const int32_t *h_pslot, int32_t scatter, void *stream);
int icv_ward_gather(icv_ward_t w, const float *d_local, int64_t ld, const int64_t *d_rows, int32_t n_rows,
                    const int32_t *d_slots, int32_t n_slots, float *out, int64_t ldo, void *stream);
int icv_ward_scatter(icv_ward_t w, float *d_local, int64_t ld, const float *v, int64_t ldv, const int32_t *h_vrow_p,
                     int32_t n_v, void *stream);
int icv_ward_scan(icv_ward_t w, const float *d_local, int64_t ld, void *stream);
int icv_ward_pack_nn(icv_ward_t w, int32_t *d_nn, float *d_dmin, void *stream);
int icv_ward_unpack_nn(icv_ward_t w, const int32_t *d_nn, const float *d_dmin, void *stream);
int icv_ward_pairs(icv_ward_t w, float *d_local, int64_t ld, int32_t all_active, int32_t *h_counts /* 4 */,
                   void *stream);
int icv_ward_round_pairs(icv_ward_t w, int32_t *h_i, int32_t *h_j);
int icv_ward_finish(icv_ward_t w, double *h_linkage, int32_t *h_rounds);

/* ---- cnv_score (tl/_scores.py:65-68): per-row sum of |x| in float64 ----------------------- */
int icv_row_abs_sum(const float *x, int64_t n_rows, int32_t n_cols, int64_t ld, double *row_sum,
                    void *stream);

/* the same on a CSR X_cnv (values float32 or float64, `indptr` n_rows + 1 int64 offsets into `data`): only the
 * stored entries are read, nothing is densified */
int icv_csr_row_abs_sum(const void *data, int32_t dtype, const int64_t *indptr, int64_t n_rows, double *row_sum,
                        void *stream);

/* per-group sums of per-cell values (cnv_score: score[g] = sum_g(row_sum) / (n_g * n_windows), tl/_scores.py:65-68):
 * sums[g] = sum of values[i] with group[i] == g (device int32 labels 0 .. n_groups-1; others are ignored), counts[g] =
 * how many.  Fixed summation order (one workgroup per group, strided partial sums + tree): the same bits for host and
 * device-resident X_cnv. */
int icv_group_sums(const double *values, const int32_t *group, int64_t n, int32_t n_groups, double *sums,
                   int64_t *counts, void *stream);

/* ---- device-resident CSR (a user-built device matrix as adata.X; X_cnv as tl.infercnv leaves it in HBM) ---------
 * icv_csr_check: the matrix is what the kernels expect -- offsets non-decreasing and inside [0, capacity] (capacity =
 * entries the index / value buffers hold), column indices inside [0, n_cols), ascending and unique within every row
 * (the host path gets this from scipy's canonical format; reference tl/_infercnv.py:115-116).  ICV_ERR_INVALID with
 * the defect in icv_last_error() otherwise.  Synchronises the stream (one flag is read back).
 * icv_csr_densify: out[q * ldo + c] = value of row rows[q] (device int64 list; NULL: rows 0 .. n_sel-1), column c, as
 * float32 -- the dense tile the fp32 MFMA contractions read (icv_corr_iqr for tl.ithcna, tl/_scores.py:197-213;
 * icv_pairwise_sqeuclidean for config 5); `data` float32 or float64. */
int icv_csr_check(const int64_t *indptr, const int32_t *indices, int64_t n_rows, int32_t n_cols, int64_t capacity,
                  void *stream);
int icv_csr_densify(const void *data, int32_t dtype, const int64_t *indptr, const int32_t *indices, const int64_t *rows,
                    int64_t n_sel, int32_t n_cols, float *out, int64_t ldo, void *stream);

/* ---- upload path of a mostly-zero DENSE host matrix (reference tl/_infercnv.py:115-116, :422-423: a dense adata.X of
 * log-counts is ~80 % zeros; PCIe is what a host-input call waits for) -- HOST functions (h_ pointers), no GPU needed:
 * icv_host_dense_row_nnz counts the stored entries (bit pattern != 0: NaN and -0.0 count) of every row of a row-major
 * float32 / float64 matrix on n_threads threads; the caller forms indptr (prefix sums, indptr[0] = 0);
 * icv_host_dense_pack writes the rows' column indices (int32, ascending) and values at indptr[r].  The arrays then cross
 * PCIe (8 or 12 bytes per stored entry instead of 4 or 8 per element) and icv_csr_scatter_dense (device pointers)
 * rebuilds the dense rows in HBM: out[r * ldo + c] = value, zeros elsewhere -- bit for bit the matrix a dense upload
 * would have delivered, so everything downstream is unchanged. */
int icv_host_dense_row_nnz(const void *h_x, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                           int64_t *h_row_nnz, int32_t n_threads);
int icv_host_dense_pack(const void *h_x, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                        const int64_t *h_indptr, int32_t *h_indices, void *h_values, int32_t n_threads);
/* The same in ONE pass over the input (what tl.infercnv uses: the host memory system is the bottleneck of the upload):
 * threads claim blocks of 16 rows in order, pack into their own staging area, take their place in the output through a
 * ticket handed on in block order and copy the block there; h_indptr (n_rows + 1) is written as well, *h_nnz = the
 * number of stored entries.  `capacity` = entries h_indices / h_values hold: ICV_ERR_NOMEM if the matrix has more
 * (h_indptr and *h_nnz are complete then; the entry arrays are not: call again with larger buffers). */
int icv_host_dense_pack_fused(const void *h_x, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                              int64_t *h_indptr, int32_t *h_indices, void *h_values, int64_t capacity,
                              int32_t n_threads, int64_t *h_nnz);
int icv_csr_scatter_dense(const void *data, int32_t dtype, const int64_t *indptr, const int32_t *indices, int64_t n_rows,
                          int32_t n_cols, void *out, int64_t ldo, void *stream);

/* ---- misc --------------------------------------------------------------------------------- */
const char *icv_last_error(void);
int icv_version(void);
/* The developer knobs (environment variables ICV_FORCE_GENERIC, ICV_NO_X16, ICV_NO_SD, ICV_WGS_PER_CU, ICV_WARD_IN_PLACE,
 * ICV_WARD_COMPACT_X, ICV_PHASE_PROFILE: kernel routes / geometries for A/B timing and kernel-against-kernel tests) are
 * read once, at the first dispatch; a process that changes them afterwards calls this to have them read again. */
void icv_developer_knobs_reload(void);
/* number of visible HIP devices (0 without a GPU); never fails */
int icv_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* INFERCNV_HIP_H */

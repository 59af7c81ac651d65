/* include/cvr_amd.h -- C ABI of libcvr_amd.so: CVR-format SpMV for MI355X (gfx950), wave64.
 *
 * This is the drop-in boundary for the ONE hot path of puckbee/CVR (/root/reference/spmv.cpp):
 *
 *   reference interface (C++ linkage, void, caller-owned arrays)        replaced by
 *   ------------------------------------------------------------------  ---------------------------
 *   readMatrix(char*, double**, int**, int**, int*, int*, int*)         cvr_mm_read / cvr_mm_free
 *       spmv.cpp:311-535, call site spmv.cpp:1771
 *   fill(double*, int)  (x = 1.0)   spmv.cpp:556-563, call :1788        cvr_fill_x
 *   pre_processing(int Nthrds, ..19 args..)                             cvr_create + cvr_preprocess
 *       spmv.cpp:565-1014, call site spmv.cpp:1857
 *   spmv_compute_kernel(..21 args.., double* h_vec, int Ntimes)         cvr_spmv (host x,y, timed) /
 *       spmv.cpp:1016-1667, call site spmv.cpp:1882                     cvr_spmv_device (async)
 *   the CSR self-check loop + verdict  spmv.cpp:1843-1850, 1916-1938    cvr_csr_spmv_host, cvr_verdict
 *   (nothing: frees are commented out, spmv.cpp:1889-1907)              cvr_destroy
 *
 * Conventions: plain C, POD structs, opaque handle, every entry point returns an int
 * status (0 = ok, <0 = error class) and never exits or throws; the message of the last error of the
 * calling thread is cvr_last_error().  The handle owns all device memory; the caller owns the host
 * CSR / x / y buffers and may free them as soon as the call that received them returns.
 * Calls on one handle must be serialised by the caller; different handles are independent.
 *
 * There is no CPU fallback behind this ABI: without a HIP device every compute entry point returns
 * CVR_ERR_NO_DEVICE.
 */
#ifndef CVR_AMD_H
#define CVR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVR_OK              0
#define CVR_ERR_INVALID    -1   /* bad argument / malformed CSR                       */
#define CVR_ERR_NO_DEVICE  -2   /* no HIP device, or device index out of range        */
#define CVR_ERR_HIP        -3   /* a HIP runtime call failed (text in cvr_last_error) */
#define CVR_ERR_IO         -4   /* loader: cannot open / not a coordinate matrix      */
#define CVR_ERR_NOMEM      -5
#define CVR_ERR_STATE      -6   /* call out of order (spmv before preprocess ...)     */
#define CVR_ERR_INTERNAL   -7   /* device converter self-check failed                 */

typedef struct cvr_handle cvr_handle;

/* Host CSR as the library reads it.  The arrays are taken LITERALLY: row r owns elements
 * row_ptr[r] .. row_ptr[r+1]-1, and col_idx indexes x directly.  The reference loader's 1-based
 * arrays (numRows+2 row pointers, columns 1..numCols) are therefore passed as a
 * matrix of numRows+1 rows and numCols+1 columns, and its "tail = nItems-1" quirk (Q9) excludes the
 * same last element the reference's CSR loop excludes. */
typedef struct {
    int64_t        nrows;
    int64_t        ncols;     /* x has ncols entries, y has nrows entries                      */
    const int64_t *row_ptr;   /* [nrows+1], non-decreasing, row_ptr[0] >= 0                    */
    const int32_t *col_idx;   /* [row_ptr[nrows]], each in [0, ncols)                          */
    const void    *vals;      /* double[] or float[] by is_f32                                 */
    int32_t        is_f32;    /* 0: fp64 values, x, y    1: fp32 values, x, y (fp32 accumulate) */
    int32_t        arrays_on_device;   /* 0: the three arrays are host memory; 1: device memory of cvr_options.device (a GPU-
                                 * resident caller): all three are checked and copied device to device; from 200 000 rows on
                                 * row_ptr never comes to the host (checked by a kernel, planned where it is), below that it
                                 * comes back once (8 B per row); the struct's size is unchanged (former padding)          */
} cvr_csr_view;

typedef struct {
    int32_t device;            /* HIP device ordinal                                           */
    int32_t steps_per_chunk;   /* S: lane-stream length of one chunk, multiple of 4; 0 = auto  */
    int64_t split_threshold;   /* rows with more remaining nnz than this may be cut at a chunk */
                               /* boundary; 0 = default (16*S; interleaved column panels: 32*S) */
    int32_t xcd_swizzle;       /* 1 (default when <0): contiguous chunk ranges per XCD; 0 off; */
                               /* 2: also consecutive chunks per CU (measured within +-2 %);   */
                               /* 3..6: runs of 2 / 4 / 8 / 16 workgroups dealt over the XCDs  */
                               /* (measured: no gain over 1)                                   */
    int32_t x_window;          /* values of x each workgroup stages in LDS with coalesced loads */
                               /* and serves its gathers from (cut to what fits the 160 KiB of */
                               /* LDS beside the row-sum stage); 0 = off (<0 = default = off)  */
    int32_t waves_per_block;   /* wavefronts (= consecutive chunks) per SpMV workgroup, 1..16; 0 = default (1).  More than */
                               /* one pays only with x_window: the chunks of a workgroup share the staged window          */
    int32_t col_panels;        /* column panels (each with its slice of x L2-resident, partial sums combined by a second
                                  kernel): 1 = off, <0 = auto (only when x is several times the L2), else the count   */
    int32_t value_dict;        /* value dictionary: one byte per slot instead of the value when the matrix has at most
                                  256 distinct values (pattern matrices); <0 = auto (default), 0 = off                */
    int32_t col_phases;        /* column phases: every chunk feeds its rows' non-zeros column range by column range (P equal
                                  ranges), so that chunks running at the same time gather from the same slice of x and that
                                  slice stays in the L2s; the sums of a row's pieces are added up in LDS, every row is still
                                  written once.  For matrices whose chunks are all resident at once and whose x is larger
                                  than an L2 (web-Google: 7.3 MB); needs ascending columns inside every row.  0 / 1 = off,
                                  <0 = auto (default)                                                                     */
    int32_t hub_table;         /* hub table: the columns with the most non-zeros (at most this many; what fits the LDS) get
                                  their x values compacted before every SpMV and staged in LDS by every workgroup (8 chunks),
                                  their gathers become ds_reads.  For power-law matrices whose x does not fit an L2.
                                  0 = off, <0 = auto (default): on when those columns hold >= 50 % of the non-zeros        */
    int32_t narrow_cols;       /* 16-bit column offsets per chunk when every chunk spans fewer than 32 767 columns (banded
                                  matrices; plain layout without value dictionary): 10 instead of 12 bytes per fp64 slot.
                                  0 = off, <0 = auto (default)                                                             */
    int32_t hub_reorder;       /* with a hub table: re-order the whole of x by column popularity before every SpMV (every column
                                  index of the image is the column's rank), so that the popular columns share cache lines.
                                  0 = off, 1 = on, <0 = auto (default): when x is at least 24 MB; never inside column panels   */
    int32_t row_tags16;        /* column phases: the chunk's row of every piece in a 16-bit tag of its own (2 more bytes per slot)
                                  instead of above the column index in the piece's last column word: lifts the limit of
                                  2^(31 - bits of ncols) rows per chunk.  0 = off, 1 = on, <0 = auto (default): when the
                                  chunks the layout wants hold more rows than the column word has room for                  */
    int32_t row_bands;         /* reserved (the 2-D form of round 3 -- row bands, each a resident launch -- was measured and not adopted:
                                  DESIGN.md 5.9).  Leave at the default (<0) or 1; values > 1 are refused with CVR_ERR_INVALID             */
    int32_t piece_max;         /* column phases: (row, phase) segments are cut into pieces of at most this many elements (at the
                                  multiples of it from the chunk's first element), so that no lane sits on one long row's
                                  segment while the others move on to the next column ranges.  A power of two (others are rounded
                                  down).  0 = whole segments,
                                  <0 = auto (default): 8 when the chunks are long enough for a lane to fall a phase behind     */
    int32_t interleave;        /* interleaved chunks: a chunk's non-zeros are dealt to the 64 lanes in COLUMN order (element e of the chunk's
                                  column-sorted list at step e / 64, lane e % 64) instead of one row per lane, so that a gather instruction
                                  reads 64 column-sorted neighbours and lanes share 128-byte lines of x; every slot carries its row, the
                                  rows' sums are accumulated in LDS (four chunks of up to 5 051 rows per workgroup).  For matrices whose x
                                  is far larger than an L2 and whose columns are scattered.  0 = off, 1 = on,
                                  <0 = auto (default): for column panels that run one per XCD and get no hub tables                       */
    int32_t gang;              /* gang chunks (interleaved images only): the chunks of a workgroup are sorted by column TOGETHER and the workgroup's
                                  wavefronts walk the common list in turn (units of two groups), adding into the chunks' accumulators in the list's
                                  order -- a token in LDS passes from unit to unit, so the sums are those of the CSR loop whatever the wavefronts'
                                  timing (bitwise reproducible) -- : four times the non-zeros share the lines of x a gather instruction touches.
                                  0 = off, 1 = on, <0 = auto (default): for interleaved column panels of four wavefronts per workgroup          */
    int32_t reserved[3];       /* 0 */
} cvr_options;
/* Automatic layout: with steps_per_chunk = 0, waves_per_block = 0, x_window < 0 and col_phases < 0 (the defaults) cvr_create
 * looks at the uploaded CSR on the device (are the rows sorted by column? which share of the non-zeros lies near the
 * diagonal?) and, for matrices whose chunks can all be resident at once, picks 6-8 chunks per workgroup sharing a 64-KiB LDS
 * window of x and/or column phases; everything else keeps one chunk per workgroup.  CVR_DEBUG=no_auto_layout in the environment
 * or any explicit value of those four options switches it off.
 * The one profiling knob is not part of this struct: CVR_DEBUG_COL_MASK in the environment (folds the gather onto a 2^k-entry table:
 * wrong results, timing only) is read by cvr_create. */

typedef struct {
    int32_t iters;
    double  mean_s, min_s, max_s;   /* per-SpMV seconds over `iters` timed launches (HIP events): the SpMV alone (compute only)  */
    double  total_s;                /* events around the whole back-to-back loop                      */
    double  h2d_s, d2h_s;           /* host<->device copies of x and y (outside mean_s)               */
    double  median_s;               /* median of the per-SpMV times                                   */
    /* rows sharded over several GPUs (cvr_spmv_multi): one step = every shard's SpMV + the all-gather of y; the slowest device
     * counts.  On one GPU there is no exchange: step_* repeat the compute figures and gather_mean_s is 0. */
    double  step_mean_s, step_min_s, step_median_s, step_max_s;
    double  gather_mean_s;          /* step_mean_s - mean_s: what the exchange adds                   */
} cvr_timing;

typedef struct {
    int64_t nrows, ncols, nnz;
    int32_t is_f32, steps_per_chunk;
    int64_t nchunks;
    int64_t nslots;            /* nchunks * 64 * S  (nnz + one pad slot per empty row + chunk tails) */
    int64_t nshared;           /* rows cut over several chunks (fix-up list length)                   */
    int64_t image_bytes;       /* device bytes of the CVR image incl. descriptors                     */
    int64_t yext_elems;        /* y_ext = [y | dump | 2 carry slots per chunk]                        */
    int64_t x_elems;           /* ncols + 1 : x_ext[ncols] must be 0 (pad slot)                       */
    double  plan_s, upload_s, convert_s;  /* planner (+ panel rule; preprocess_fused: the whole chain up to the converter's end), H2D of the CSR,
                                           * device time of segment table + conversion (preprocess_fused: from the planner's first kernel) */
    int32_t col_panels;        /* 1, or the number of column panels the matrix was cut into                          */
    int32_t value_dict;        /* 0, or the number of dictionary entries (distinct values + the pad slots' 0)         */
    int32_t col_phases;        /* 1, or the number of column phases                                                    */
    int32_t waves_per_block;   /* wavefronts (chunks) per SpMV workgroup                                                */
    int32_t x_window;          /* values of x every workgroup stages in LDS (0 = none)                                  */
    int32_t lds_bytes;         /* dynamic LDS of one SpMV workgroup                                                     */
    int64_t nsegments;         /* column phases: (row, phase) segments over all chunks (0 otherwise)                    */
    int64_t chunk_row_cap;     /* column phases: most rows the planner gives a chunk (their sums live in LDS); 0 = none */
    double  near_diagonal_share;   /* automatic layout: share of the non-zeros within a quarter window of the diagonal (0 if not probed) */
    int32_t hub_entries;           /* hub table: columns staged in LDS (0 = none)                                          */
    int32_t narrow_cols;           /* 1: the image stores 16-bit column offsets (narrow chunks)                             */
    int32_t hub_reorder;           /* 1: the image's column indices are popularity ranks, x is re-ordered before every SpMV  */
    int32_t row_tags16;            /* 1: the image carries 16-bit row tags (column phases with long chunks / wide matrices)   */
    double  hub_share;             /* share of the non-zeros in the hub columns that were (or could have been) chosen      */
    double  hub_select_s;          /* the device pass that counted and ranked the columns (0 if not run)                  */
    double  probe_s;               /* automatic layout: the device pass over the CSR (sortedness, near-diagonal share), 0 if not run */
    double  dict_s;                /* the value-dictionary detection pass over the uploaded values (part of upload_s) */
    double  preprocess_wall_s;     /* host wall time of cvr_preprocess: convert_s (device events) + its temporary allocations and the final sync */
    int32_t row_bands;             /* 1, or the number of row bands (resident launches per SpMV)                                           */
    int32_t piece_max;             /* column phases: longest piece of a lane stream (0 = whole (row, phase) segments)                     */
    int32_t spmv_launches;         /* launches of the SpMV kernel one SpMV is made of: 1 (also with column panels that run one per XCD at a
                                      time: all rounds of eight share one grid), or one per panel when each panel runs over the whole chip;
                                      combine / fix-up / hub kernels not counted                                                            */
    int32_t preprocess_fused;      /* 1: cvr_create ran analysis, chunk plan, segment table and conversion as one submission (resident
                                    * layouts, cvr_fused.hip: plan_s covers all of it, the first cvr_preprocess has nothing left to do) */
    int32_t interleave;            /* 1: the image's chunks are interleaved (cvr_options.interleave)                                      */
    int32_t gang;                  /* > 0: gang chunks (cvr_options.gang) -- the wavefronts of a workgroup that walk one common list            */
} cvr_info;

void        cvr_default_options(cvr_options *opt);
const char *cvr_last_error(void);
const char *cvr_version(void);
int         cvr_device_count(void);                     /* 0 when there is no usable HIP device */

/* ---- the handle: one matrix (or one row shard of it) on one GPU ------------------------------ */
/* Validates the CSR, uploads it, looks at it on the device to choose the layout, plans the chunks (on the device from
 * 200 000 rows on, else on the host: the same plan).  (pre_processing's setup half: chunk partition + row search,
 * spmv.cpp:584-694.)  For a single image in the resident layout the tracker loop follows in the same submission of kernels
 * (cvr_info.preprocess_fused): the handle comes back converted, and cvr_preprocess only releases the CSR and reports the times. */
int cvr_create(cvr_handle **out, const cvr_csr_view *csr, const cvr_options *opt);
/* CSR -> CVR64 on the device (the tracker loop, spmv.cpp:711-1000); `seconds` = what the reference
 * prints at spmv.cpp:1009.  Frees the device copy of the CSR unless keep_csr != 0. */
int cvr_preprocess(cvr_handle *h, int keep_csr, double *seconds);
int cvr_get_info(const cvr_handle *h, cvr_info *info);
int cvr_destroy(cvr_handle *h);

/* y = A x.  x_host: ncols values, y_host: nrows values (type by is_f32).  One untimed warm-up launch,
 * then `iters` timed launches; y of the last one is copied back.  (spmv.cpp:1016-1667; unlike the
 * reference, spmv.cpp:1026-1033, nothing the result needs is left outside the timed region.) */
int cvr_spmv(cvr_handle *h, const void *x_host, void *y_host, int iters, cvr_timing *timing);

/* Asynchronous single SpMV on caller-provided device buffers and stream (a hipStream_t passed as
 * void*; NULL is HIP's null stream, cvr_stream(h) is the handle's own).  x_dev must hold info.x_elems values with
 * x_dev[ncols] == 0; y_dev must hold info.yext_elems values (the first nrows are y; the rest is
 * scratch: carry slots of rows cut over chunks).  Rows without
 * non-zeros are written as 0 on every call; y needs no zeroing.  The call makes the handle's device current.  A handle with
 * column panels (info.col_panels > 1) keeps the panels' partial sums in one buffer of its own: launches of such a handle on
 * different streams are ordered one after the other by the library (an event recorded on the stream it leaves when a launch
 * comes on another one; the stream of the earlier launches must still exist then, or the library waits for the device), so they
 * do not overlap.  Launches on one stream are ordered by that stream alone. */
int cvr_spmv_device(cvr_handle *h, const void *x_dev, void *y_dev, void *stream);
/* the same, `n` launches back to back (the Ntimes loop of spmv.cpp:1024 without a host round trip per launch) */
int cvr_spmv_device_repeat(cvr_handle *h, const void *x_dev, void *y_dev, void *stream, int n);
/* The column-panel count cvr_create chooses for col_panels = -1 (host only, no device needed): 1 unless x is >= 24 MB
 * -- or >= 12 MB and the matrix is too large for the resident layout (more slots or rows than its workgroups hold in one pass) --
 * and the estimated share of x gathers missing a 4-MiB L2 (*l2_miss_estimate, sampled over eight windows of 65 536
 * rows) exceeds 0.17; then one panel per 1.8 MB of missing x, counted in rounds of eight (one panel per XCD at a time: eight up
 * to 27 MB of x, then 8 * ceil(x / 20.8 MB)); doubled once for thin lists (fewer than two non-zeros of a workgroup's list per
 * 128-byte line of a panel's slice of x, and fewer than 0.15 (row, panel) pairs per non-zero).  cvr_create differs from this host
 * rule in what it measures on the device: hub tables (popular columns) widen or drop the panels, unevenly filled panels are
 * doubled, and from 8 MB of x on a matrix whose non-zeros are not near the diagonal is asked the panel question as well.
 * Returns the count (>= 1) or a negative error. */
int cvr_auto_panels(const cvr_csr_view *csr, double *l2_miss_estimate);

/* Optional tuning of steps_per_chunk by measurement: builds the matrix with S = 8, 12, ... 64 on the device, times the
 * SpMV of each (about 1.5 ms of launches per candidate) and returns the fastest; the caller then passes it as
 * cvr_options.steps_per_chunk to cvr_create.  The default rule (steps_per_chunk = 0) needs no tuning on matrices
 * that fill the GPU many times over; on small ones (a row shard of web-Google on one of 8 GPUs) which chunk counts run
 * fastest depends on how the workgroups fall onto the CUs, and measuring beats the rule by 10-20 %.
 * Host arrays are uploaded once and every candidate is built from the device copy; *tuning_s counts as preprocessing time. */
int cvr_tune_steps(const cvr_csr_view *csr, const cvr_options *opt, int32_t *best_steps, double *best_spmv_s, double *tuning_s);
/* The same over the whole layout: besides S = 8 .. 64 with one chunk per workgroup it measures, for matrices small enough,
 * the resident layout (8 or 4 chunks per workgroup, every workgroup on its own CU at once) with and without a 64-KiB LDS
 * window of x and with and without column phases, and returns the fastest set of options in *best (pass it to cvr_create). */
int cvr_tune(const cvr_csr_view *csr, const cvr_options *opt, cvr_options *best, double *best_spmv_s, double *tuning_s);

/* ---- one call = all GPUs of the process ------------------------------------------------------------------
 * In the reference ONE call drives all threads (pre_processing spmv.cpp:1857 -> omp parallel num_threads at :577;
 * spmv_compute_kernel :1882 -> :1034).  The multi-device handle is that for GPUs: it cuts the rows into one contiguous block
 * per device with balanced predicted time (cvr_row_partition_cost: non-zeros plus a cost per row; binary search on row_ptr, the reference's per-thread trick of
 * spmv.cpp:631-667, but at row boundaries, so no row spans devices and nothing is reduced across them), builds one shard
 * handle per device (x replicated), owns the device vectors, the communicators (ncclCommInitAll) and the all-gather of y.
 * A device may be listed several times (CVR_DEVICES=0,0,0 on a one-GPU box): sharding, handles and gather layout stay,
 * device-to-device copies stand in for RCCL, which needs distinct devices. */
typedef struct cvr_multi cvr_multi;
/* bounds[nparts + 1]: rows [bounds[p], bounds[p+1]) go to part p; returns the largest part's row count (>= 0) or < 0 on error.
 * Host only, no device needed.  The one partition rule of this library (the host program, bench.py and cvr_amd/shard.py use it). */
int64_t cvr_row_partition(int64_t nrows, const int64_t *row_ptr, int32_t nparts, int64_t *bounds);
/* The same cut on predicted TIME instead of non-zeros: a row costs its non-zeros plus row_cost_milli / 1000 of a non-zero (its hand-out,
 * its accumulator, its store: a fit over the eight row shards of R-MAT-26, whose kernels ran 851 .. 1 074 us at equal non-zeros --
 * t = 6.33 ns per 1 000 non-zeros + 7.7 ns per 1 000 rows, i.e. 1.22 non-zeros per row; profiles/r03_rank_emulation_rmat26.json).
 * bounds[p] = the first row r with 1000 * nnz(rows before r) + row_cost_milli * r >= 1000 * floor(nnz * p / nparts) + floor(nrows * row_cost_milli * p / nparts).  row_cost_milli = 0 is cvr_row_partition (the
 * reference balances non-zeros only, spmv.cpp:584-627); CVR_ROW_COST_MILLI_DEFAULT is what cvr_create_multi, spmv.cvr and bench.py use
 * (spmv.cvr: CVR_PARTITION=nnz in the environment keeps the reference's rule). */
#define CVR_ROW_COST_MILLI_DEFAULT 1250
int64_t cvr_row_partition_cost(int64_t nrows, const int64_t *row_ptr, int32_t nparts, int32_t row_cost_milli, int64_t *bounds);
/* csr: host arrays of the whole matrix; opt: as for cvr_create (opt->device is ignored); devices[ndevices]: HIP ordinals */
int cvr_create_multi(cvr_multi **out, const cvr_csr_view *csr, const cvr_options *opt, const int32_t *devices, int32_t ndevices);
int cvr_preprocess_multi(cvr_multi *m, int keep_csr, double *seconds);      /* seconds: the slowest shard's conversion + planning */
/* y = A x through host buffers: x is replicated to every device, every shard computes its rows, the y slices are all-gathered
 * (every device ends up with the whole y), y comes back from the first device's gathered copy.  `iters` timed steps
 * compute-only, then `iters` with the gather (timing->mean_s ... and timing->step_*). */
int cvr_spmv_multi(cvr_multi *m, const void *x_host, void *y_host, int iters, cvr_timing *timing);
int cvr_multi_shards(const cvr_multi *m);                                    /* number of shards (= ndevices) */
int cvr_multi_info(const cvr_multi *m, int32_t shard, cvr_info *info, int64_t *row_begin, int64_t *row_end, int32_t *device);
int cvr_multi_uses_rccl(const cvr_multi *m);                                 /* 1: ncclAllGather; 0: device-to-device copies (or one shard) */
/* shard handles that exist already (loaded from image caches, or built by the caller): adopted by a multi handle, which then owns
 * them; bounds[n + 1], devices[n]; handle g holds rows [bounds[g], bounds[g+1]) and is preprocessed */
int cvr_multi_from_handles(cvr_multi **out, cvr_handle **shards, const int64_t *bounds, const int32_t *devices, int32_t n);
cvr_handle *cvr_multi_handle(cvr_multi *m, int32_t shard);                   /* shard's handle (owned by m), e.g. for cvr_save_image */
int cvr_destroy_multi(cvr_multi *m);

/* ---- rows sharded over GPUs, one process per GPU: the exchange step ------------------------------------
 * The reference's threads share one y in host memory (spmv.cpp:1280-1282, 1640-1649); with one row shard per GPU
 * (contiguous rows, cut at row boundaries, x replicated) the shards' y slices are all-gathered over RCCL / xGMI.
 * RCCL is loaded on first use (dlopen; the instance already in the process, e.g. PyTorch's, is preferred), so
 * single-GPU users never pay for it. */
typedef struct cvr_comm cvr_comm;
#define CVR_COMM_ID_BYTES 128
/* rank 0 calls this and hands the 128 bytes to every rank by its own means (file, socket, torch.distributed) */
int cvr_comm_unique_id(void *id128);
/* collective over all ranks; `device` is this rank's GPU */
int cvr_comm_create(cvr_comm **comm, const void *id128, int nranks, int rank, int device);
int cvr_comm_destroy(cvr_comm *comm);
/* What RCCL itself says about the communicator -- ncclCommCount, ncclCommUserRank, ncclGetVersion (-1 where the loaded library lacks the call):
 * a record of a multi-GPU run can show that the collective saw N ranks.  Any pointer may be NULL. */
int cvr_comm_info(cvr_comm *comm, int *nranks, int *rank, int *rccl_version);
/* one all-gather of `count` values per rank (type by is_f32) on `stream`: recv_dev holds nranks * count values */
int cvr_comm_all_gather(cvr_comm *comm, const void *send_dev, void *recv_dev, int64_t count, int is_f32, void *stream);
/* `n` sharded SpMVs of the fixed-x loop (spmv.cpp:1024), each followed by the all-gather of this rank's y slice
 * (the first max_rows values of y; every rank passes the same max_rows >= its row count).  Step k computes into
 * y_dev[k & 1] and gathers into yall_dev[k & 1] (nranks * max_rows values).  overlap = 0: SpMV and gather follow
 * each other on `stream` (two enqueues per step; the cheapest for the host).  overlap = 1: the gather runs on the
 * communicator's own stream, so the gather of step k overlaps the SpMV of step k + 1, and buffers are reused only
 * after the gather that used them has finished (two events per step: worth it when the gather outlasts the
 * SpMV by more than their cost).  Every rank must pass the same n and overlap.  On return `stream` is ordered after
 * every gather; *last_buf = index of the buffers holding the last step.  y_dev[i] must hold
 * max(info.yext_elems, max_rows) values.  No host synchronisation. */
int cvr_spmv_gather_repeat(cvr_handle *h, cvr_comm *comm, const void *x_dev, void *const y_dev[2], void *const yall_dev[2],
                           int64_t max_rows, int n, int overlap, void *stream, int *last_buf);

/* The iterative caller (power iteration): x <- A x / ||A x||, `iters` times, everything on the device and on `stream`
 * with no host round trip inside the loop; dot products use a fixed reduction tree, so results are bitwise reproducible.
 * x_dev: info.x_elems values, in: the start vector (any non-zero), out: the normalised iterate; x_dev[ncols] stays 0.
 * *lambda = x_k . (A x_k) / x_k . x_k of the last iteration (Rayleigh quotient).  Inside the loop a step's dot products and
 * the scaling of the next x are one pass over the vectors: x is scaled by the norm of the step before (||x|| stays between
 * 1/lambda and lambda), the last iterate is normalised exactly.  comm = NULL: one GPU, the handle holds the whole
 * square matrix.  comm != NULL: the handle holds this rank's row block of a square matrix of ncols rows, bounds[nranks+1]
 * are the row offsets of all blocks, and every iteration all-gathers y over RCCL and rebuilds the replicated x from
 * it -- the one setting where the exchange step is on the critical path.  On one GPU with an image of the resident layout (column
 * phases, no row cut over chunks) the step's dot products and the next iterate come out of the SpMV kernel's write-out: one launch per
 * iteration (web-Google shape: 26 us per iteration, SpMV alone 21).  Synchronises `stream` before returning.
 * (The reference has no such loop: its Ntimes loop, spmv.cpp:1024, recomputes one y.) */
int cvr_power_iteration(cvr_handle *h, cvr_comm *comm, const int64_t *bounds, int iters, void *x_dev, double *lambda,
                        double *seconds_per_iter, void *stream);

/* the handle's own device vectors (valid until cvr_destroy) and stream */
void *cvr_x_device(cvr_handle *h);
void *cvr_y_device(cvr_handle *h);
void *cvr_stream(cvr_handle *h);
/* `iters` back-to-back launches on the handle's stream and buffers between two HIP events;
 * returns mean seconds per launch.  No host copies.  (bench.py's roofline leg) */
int cvr_spmv_bench(cvr_handle *h, int warmup, int iters, double *mean_s);
/* Diagnostics (no counterpart in the reference, which times its phases with microtime(), spmv.cpp:575 / 1009, 1033 / 1656): a handle
 * created with CVR_DEBUG=phase_clocks in the environment (headline layout: fp64, column phases + window + dictionary) runs the SpMV
 * kernel in a build that stamps the chip's 100-MHz real-time counter per wavefront at entry / prologue done / window barrier passed /
 * loop done / rows stored; this copies the stamps of the last SpMV out: [workgroup][16 wavefronts][8] =
 * {t_entry, t_prologue, t_window (barrier passed), t_loop_end, t_stored, XCC id, t_arrived at the window barrier (loaders: hardware id), kind (1 computing, 2 loader, 0 none)}.
 * *nwords = words available; tools/phase_clocks.py makes the per-XCD histogram. */
int cvr_debug_phase_clocks(cvr_handle *h, unsigned long long *out, int64_t max_words, int64_t *nwords);

/* Calibration for the roofline: a 16-byte-per-lane copy kernel over `bytes` of device memory (read
 * `bytes`, write `bytes`); returns the mean read+write rate in GB/s over `iters` launches on `device`. */
int cvr_device_copy_bench(int device, int64_t bytes, int iters, double *gbs);

/* Copies the device-resident CVR64 image back for inspection (tests compare it bit for bit with the
 * CPU mirror).  Any pointer may be NULL.  Sizes: cols_vals = image_bytes of the stream part
 * (nchunks * S/4 * group_bytes), desc = 4 u32 per chunk, target = 64 u8 per chunk,
 * shared = 3 i64 per shared row {row, first chunk, last chunk}.  With column phases desc[k][1] counts the (row, phase)
 * segments of the chunk and the last column word of a segment carries the chunk's row of the segment above the column index. */
int cvr_export_image(cvr_handle *h, void *stream_image, uint32_t *desc, uint8_t *target, int64_t *shared);
/* The rest of an image with gang chunks (cvr_options.gang), for the same comparison: group_first_cols = nchunks * S/4 u32 -- the first column of
 * every group of every gang, gang b's groups from (b * waves_per_block) * S/4 on, zeros behind its last (all zeros when the image carries 16-bit
 * tags) --, desc2 = 2 u32 per chunk {groups of the gang that hold non-zeros (at its first chunk; 0 at the others), rows of the chunk}.
 * Either pointer may be NULL.  CVR_ERR_STATE for images without gang chunks. */
int cvr_export_gang(cvr_handle *h, uint32_t *group_first_cols, uint32_t *desc2);
/* host planner only (no device needed): chunk boundaries for a row_ptr; returns nchunks or <0.
 * out arrays (each may be NULL) need room for cvr_plan_bound(nrows, nnz, S) chunks. */
int64_t cvr_plan_bound(int64_t nrows, int64_t nnz, int32_t S);
int64_t cvr_plan_chunks(int64_t nrows, const int64_t *row_ptr, int32_t S, int64_t split_threshold,
                        int64_t *nz_begin /*[n+1]*/, int64_t *row_first, int64_t *nseg, int64_t *pad_cnt);

/* diagnostics (needs a device): plans row_ptr with the device planner (cvr_plan_dev.hip) and with the host planner and
 * compares the two plans field by field (CVR_OK = identical); seconds of both, chunk count. */
int cvr_plan_selfcheck(int device, int64_t nrows, const int64_t *row_ptr, int32_t S, int64_t split_threshold, int64_t max_rows,
                       double *host_seconds, double *device_seconds, int64_t *nchunks);

/* ---- host side of the reference program ------------------------------------------------------ */
#define CVR_MM_REFCOMPAT 0   /* the reference loader's arrays bit for bit (quirks Q1-Q9)           */
#define CVR_MM_STRICT    1   /* Matrix-Market semantics: 0-based, fp64 values, no padding, 64-bit  */
typedef struct {
    int64_t  nrows, ncols, nnz;   /* of the arrays below, taken literally (see cvr_csr_view)      */
    int64_t  ref_numRows, ref_numCols, ref_nItems, ref_nItemsRaw;  /* header rows/cols, padded and */
                                                                   /* raw entry counts (refcompat) */
    int64_t *row_ptr;
    int32_t *col_idx;
    double  *vals;
} cvr_mm_matrix;
int  cvr_mm_read(const char *path, int mode, cvr_mm_matrix *out);   /* readMatrix, spmv.cpp:311-535 */
void cvr_mm_free(cvr_mm_matrix *m);
/* binary image of a parsed matrix (all fields of cvr_mm_matrix): skips the text parse on the next run */
int  cvr_mm_write_bin(const char *path, const cvr_mm_matrix *m);
int  cvr_mm_read_bin(const char *path, cvr_mm_matrix *out);
/* The same, keyed to the source: the identity of a file is its size, its modification time (ns) and a 64-bit FNV-1a hash of its
 * first and last MiB, together with the loader mode.  A keyed image is only read back under the key it was written with:
 * CVR_ERR_STATE says the source has changed since (or the image carries no key) and the caller parses the text again. */
typedef struct { int64_t size, mtime_ns; uint64_t hash; int32_t mode, reserved; } cvr_source_key;
int  cvr_source_key_of(const char *path, int mode, cvr_source_key *key);
int  cvr_mm_write_bin_keyed(const char *path, const cvr_mm_matrix *m, const cvr_source_key *key);
int  cvr_mm_read_bin_keyed(const char *path, const cvr_source_key *expect, cvr_mm_matrix *out);
/* readMatrix through the cache beside the file (<mtx>.ref.csrbin / <mtx>.strict.csrbin): the cache when its key is the file's,
 * else the text, after which the cache is rewritten; *cache_hit (may be NULL) says which */
int  cvr_mm_read_cached(const char *mtx_path, int mode, cvr_mm_matrix *out, int *cache_hit);
/* The converted CVR64 image of a handle on disk (after cvr_preprocess; single images, column panels, hub tables alike): a second run
 * on the same matrix loads it straight into device memory and skips analysis, planner and converter (the reference repeats its
 * pre_processing on every run, spmv.cpp:1857).  The file is keyed by the source file's identity (`key`; NULL = none), the options,
 * the device's CU / XCD counts, the format and library version; cvr_load_image returns CVR_ERR_STATE when any of them differs
 * (the caller then runs cvr_create + cvr_preprocess and saves again).  The loaded handle computes the same y, bit for bit.
 * opt: the options the image must have been built with (NULL = defaults; opt->device = where to load it).  *seconds: load time. */
int  cvr_save_image(cvr_handle *h, const char *path, const cvr_source_key *key);
int  cvr_load_image(cvr_handle **out, const char *path, const cvr_source_key *expect, const cvr_options *opt, double *seconds);
/* x[j] = 1.0 (mode 0; fill, spmv.cpp:556-563) or splitmix64(0xC0FFEE, j) -> [-1,1) (mode 1) */
void cvr_fill_x(double *x, int64_t n, int mode);
/* the reference's self-check loop, OpenMP over rows, j ascending (spmv.cpp:1843-1850) */
void cvr_csr_spmv_host(int64_t nrows, const int64_t *row_ptr, const int32_t *col_idx, const double *vals,
                       const double *x, double *y, int nthreads);
/* rows i in [0, nrows_checked) with (y[i]-yref[i])^2 > 1e-6 (spmv.cpp:1916-1929) */
int64_t cvr_verdict(const double *y, const double *yref, int64_t nrows_checked);

#ifdef __cplusplus
}
#endif
#endif

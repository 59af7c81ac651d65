// cvr_kernels.h -- launch wrappers of the gfx950 kernels (cvr_convert.hip, cvr_spmv.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cvr_format.h"

namespace cvr {

// diagnostics (cvr_debug.cpp): CVR_DEBUG="name[=value],..." in the environment, parsed by debug_refresh() where a call of the C ABI begins;
// debug_env(name) = the value ("1" without one) or nullptr
void        debug_refresh();
const char *debug_env(const char *name);

struct DeviceImage {
    int32_t  S = 0, G = 0;
    bool     f32 = false;
    uint32_t nchunks = 0;
    uint32_t nrows = 0;          // y rows; y_ext[nrows] is the dump slot
    uint32_t pad_col = 0;        // = ncols: x_ext[pad_col] == 0
    uint8_t *stream = nullptr;   // nchunks * G * group_bytes
    uint4   *desc = nullptr;     // [nchunks] {row_first, nseg, head_dest, last_dest}
    uint8_t *target = nullptr;   // [nchunks][64]
    int64_t *shared = nullptr;   // [nshared][3] {row, c0, c1}
    uint32_t nshared = 0;
    uint32_t ncus = 256;            // CUs of the device (hipDeviceProp_t::multiProcessorCount)
    int      xcd_swizzle = 1;       // 0 off, 1 contiguous chunk range per XCD, 2 additionally consecutive chunks per CU (experiment)
    uint32_t ystage = 1024;         // row sums a wavefront stages in LDS (multiple of 64, <= kYStageMax)
    const void *dict = nullptr;     // value dictionary: ndict values of T sorted by bit pattern (device), or null
    uint32_t  ndict = 0;
    uint32_t wpb = 1;               // wavefronts (= consecutive chunks) per SpMV workgroup, 1..kMaxWavesPerBlock
    uint32_t *win_base = nullptr;   // [ceil(nchunks / wpb)] first column of each workgroup's LDS window of x
    uint32_t win_elems = 0;         // window length in values (0 = no window)
    uint32_t col_mask = kColMask;   // profiling only: a narrower mask folds the x gather onto a small table
    // column phases (cvr_options.col_phases > 1): a chunk feeds its rows' pieces phase by phase (column range by column
    // range), every (row, phase) pair with a non-zero is one segment; the segments' sums are added into per-row
    // accumulators in LDS and every row is written once at the end of the chunk
    uint32_t  phases = 1;
    uint32_t  phase_width = 0;      // columns per phase (multiple of 16)
    uint2    *desc2 = nullptr;      // [nchunks] {first entry of the chunk in the conversion-time segment table, rows with a segment in the chunk}
    bool      tag16 = false;        // column phases: the rows of the pieces stand in 16-bit tags of their own (col_bits = 31)
    uint32_t  col_base = 0;         // interleaved column panels: the image's column indices are relative to this column (pad_col = the panel's width)
    bool      ilv = false;          // interleaved chunks (cvr_ilv.hip): the image is written in the column-phase format with every slot a piece of its own
    uint32_t  gang = 0;             // > 0 (= wpb): gang chunks -- the wpb chunks of a workgroup sorted together, walked by its wavefronts in turn (cvr_format.h)
    uint32_t *gbase = nullptr;      // gang chunks without 16-bit tags: [nchunks * G] the first column of every group (group g of gang b at (b * wpb) * G + g)
    unsigned long long *prof = nullptr;      // CVR_DEBUG=phase_clocks: [workgroups * 16 wavefronts][8] time stamps of spmv_seg_kernel's phases (diagnostics; null otherwise)
    uint32_t  prof_words = 0;
    uint32_t  ilv_helpers = 0, ilv_ahead = 16, ilv_per_line = 1;
    uint32_t  stream_mod = 0;       // profiling only (CVR_DEBUG=stream_mod=M): every chunk streams the image of chunk k % M (L2-resident stream, wrong sums)
    uint32_t  ilv_flip = 0;         // interleaved: 1 = every other SpMV walks the workgroups in reverse order (what the last one streamed last is still in the Infinity Cache); set per launch
    uint32_t  ilv_stream_nt = 0; // interleaved images and images with a hub table: the stream is loaded non-temporally (images that do not stay in the caches between SpMVs: ilv_runtime_settings)
    uint32_t  flip_now = 0;      // interleaved: helper wavefronts per chunk (scalar prefetch of the stream), how many groups ahead, loads per line
    uint32_t  piece_max = 0;        // column phases: (row, phase) segments are cut into pieces at the multiples of this many elements from the chunk's first (0 = whole segments)
    uint32_t  col_bits = 31;        // column phases: the LAST column word of a segment carries the chunk's row of the segment
                                    // above the column index: bits [col_bits, 31); bit 31 stays the end flag
    // hub table (cvr_hub.hip): the hub_n columns with the most non-zeros; a slot of such a column holds its table index and
    // kHubBit; the workgroup stages hub_x[0 .. hub_n) (= x[hub_cols], compacted before every SpMV) in LDS in front of the window
    bool      c16 = false;          // narrow chunks: 16-bit column offsets from cbase[k] (plain layout without dictionary only)
    uint32_t *cbase = nullptr;      // [nchunks] smallest column of the chunk
    uint32_t  hub_n = 0;
    uint32_t  order_n = 0;          // > 0 (= ncols): every column index of the image is the column's popularity rank; hub_x holds the whole re-ordered x
    int32_t  *hub_cols = nullptr;   // [hub_n] the hub columns, by non-zeros descending
    int32_t  *hub_index = nullptr;  // [ncols] table index of a column, -1 = none (conversion only)
    uint32_t *hub_bitmap = nullptr; // [ncols / 32] bit c = column c is a hub (conversion only: saves the lookup of the cold columns)
    void     *hub_x = nullptr;      // [hub_n rounded up to 4] values of T
};

struct HubSelection { int32_t *hub_cols = nullptr, *hub_index = nullptr; uint32_t *hub_bitmap = nullptr; uint32_t H = 0, order_n = 0; double share = 0; };
// columns are ranked on every stride-th non-zero: a sample of at most 2^23
inline int64_t hub_sample_stride(int64_t nnz) { return nnz > (int64_t)(1 << 23) ? (nnz + (1 << 23) - 1) >> 23 : 1; }
// the (at most hmax) columns with the most non-zeros among col_idx[n0, n1), at least 2 each; share = their part of the non-zeros; synchronises st
hipError_t select_hubs(const int32_t *ci, int64_t n0, int64_t n1, int64_t ncols, uint32_t hmax, HubSelection *out, hipStream_t st, bool full_order = false);
void       free_hubs(HubSelection &s);
// select_hubs' share alone (no ranking, no tables): synchronises st
// Scratch: device memory the caller lends for the duration of the call (round 5: the analysis passes of cvr_create allocated and freed their own -- a
// hipMalloc + hipFree pair costs more than the kernels between them); too small or null: the pass allocates.
struct Scratch { void *p = nullptr; size_t bytes = 0; };
// col_share (optional, kColBins entries): the share of the sampled non-zeros in each of kColBins equal column ranges
constexpr uint32_t kColBins = 1024;
hipError_t hub_share_device(const int32_t *ci, int64_t n0, int64_t n1, int64_t ncols, uint32_t hmax, double *share, hipStream_t st, Scratch lent = Scratch(), double *col_share = nullptr);
hipError_t launch_hub_gather(const DeviceImage &img, const void *x_ext, hipStream_t st);     // hub_x = x[hub_cols]

struct DeviceCsr {
    const int64_t  *row_ptr = nullptr;
    const int32_t  *col_idx = nullptr;
    const void     *vals = nullptr;
    const int64_t  *nz_begin = nullptr;   // [nchunks+1]
    const uint32_t *pad_cnt = nullptr;    // [nchunks]
    const uint8_t  *codes = nullptr;      // optional: the dictionary code of every value, same indexing as vals (launch_dict_codes)
};

// column phases: the segment table of every chunk, built on the device from the CSR and the plan
struct SegTable {
    uint32_t *cnt = nullptr;        // [nchunks + 1] segments per chunk; [nchunks] = their sum
    int64_t  *begin = nullptr;      // [nchunks * 64 S] chunk k's segments at k * 64 S: first CSR element of the segment, -1 = pad slots
    uint32_t *len = nullptr;        // same layout: slots of the segment
    uint16_t *row = nullptr;        // same layout: the chunk's row of the segment (rows-in-chunk = the pad segment's dump entry)
    uint32_t *flags = nullptr;      // [2]: [0] bit 0 = a row's columns are not ascending
    bool      packed = false;       // set by launch_seg_build: the entries are 8-byte records {begin - chunk start | length << 16, row} in `begin`'s
                                    // memory (begin 0xffff = pad slots; chunks of fewer than 65 535 slots): one store and one load per segment
};
// the segment table of every chunk (one workgroup each: counts per phase, then (begin, length, row) in (phase, row) order); writes
// desc[k].y = segments of chunk k, desc2[k].x = where they start, st.cnt, st.flags
// (nchunks_dev, here and below: the number of chunks is read on the device -- a plan that has not come back to the host -- and
// img.nchunks is the room the launch is made for)
hipError_t launch_seg_build(const DeviceImage &img, const DeviceCsr &csr, SegTable &st, hipStream_t st_, const uint32_t *nchunks_dev = nullptr, bool with_total = true);
hipError_t launch_seg_total(const SegTable &st, uint32_t nchunks, const uint32_t *nchunks_dev, uint32_t *total_out, hipStream_t s);      // *total_out = sum of st.cnt
bool       seg_table_packed_ok(const DeviceImage &img);      // launch_seg_build will write packed entries (st.len / st.row are not needed)

// CSR -> CVR64 (one wavefront per chunk).  *err_flag (device u32, zeroed by the caller) gets bit 0 if a
// lane stream did not drain, bit 1 if stealing found no over-full lane, bit 2 if a value is missing from the dictionary.
hipError_t launch_convert(const DeviceImage &img, const DeviceCsr &csr, uint32_t *err_flag, hipStream_t st, const SegTable *seg = nullptr, const uint32_t *nchunks_dev = nullptr);
// interleaved chunks (cvr_ilv.hip): the images' non-zeros sorted by column inside every chunk and dealt to the lanes in that order; the
// images of one handle (same chunk length, value type, dictionary, tag width) go together: one sort, one writing pass.
// `scratch`: convert_interleaved_scratch(sum of the non-zeros, ..) bytes of device memory; *err_flag bit 2: a value that is not in the dictionary
size_t     convert_interleaved_scratch(int64_t nnz, uint32_t nchunks, bool gang = false);      // gang: + the device-wide sort's keys, positions and storage (cvr_ilv.hip)
hipError_t launch_convert_interleaved(const DeviceImage *const *imgs, const DeviceCsr *csrs, const int64_t *n0, const int64_t *n1, int n, uint32_t *err_flag, void *scratch,
                                      size_t scratch_bytes, hipStream_t st);
// codes[j] = dictionary code of vals[j], j in [n0, n1) (one coalesced pass; *err_flag bit 2: a value that is not in the dictionary)
hipError_t launch_dict_codes(const void *vals, int64_t n0, int64_t n1, bool f32, const void *dict, uint32_t ndict, uint8_t *codes, uint32_t *err_flag, hipStream_t st);

// value-dictionary detection over vals[n0, n1) on the device: `table` = 1024 u64 slots preset to all ones, flags[0] bit 0 =
// more than kDictMax distinct values, bit 1 = the all-ones pattern occurs, flags[1] = entries in the table
// ---- column-panel split on the device (cvr_split.hip), for CSR arrays that are already there ----
constexpr int kMaxSplitPanels = 64;
struct DeviceSplit {
    int32_t  *ci = nullptr;      // [nnz] column indices, panel after panel, original order inside a panel
    void     *va = nullptr;      // [nnz] values, same order
    uint32_t *rows = nullptr;    // [nsub] the row of every sub-row, panel after panel (ascending inside a panel)
    int64_t  *rp = nullptr;      // [nsub + 1] where every sub-row starts in ci / va (positions over all panels); rp[nsub] = nnz
    int64_t   nnz = 0, nsub = 0;
    int64_t   off[kMaxSplitPanels + 1];    // host: first non-zero of every panel, off[P] = nnz
    int64_t   sub0[kMaxSplitPanels + 1];   // host: first sub-row of every panel, sub0[P] = nsub
};
// panels = columns [p * width, (p + 1) * width); needs nnz, nrows < 2^32; synchronises `st`
hipError_t split_panels_device(const int64_t *rp_dev, const int32_t *ci_dev, const void *va_dev, bool f32, int64_t nrows, int64_t nz0,
                               int64_t nz1, int64_t width, int P, DeviceSplit *out, hipStream_t st);
void       free_device_split(DeviceSplit &s);
// the panel rule's second question (cvr_split.hip): (row, panel) pairs of the same windows for panels of `width` columns
// (refs[w] = the window's non-zeros)
hipError_t panel_pairs_device(const int64_t *rp_dev, const int32_t *ci_dev, const int64_t *r0_host, int nwin, int64_t W, int64_t width, double *pairs, double *refs, hipStream_t st,
                              Scratch lent = Scratch());
// the panel rule's L2 model for a device-resident CSR (cvr_split.hip): per window of W rows, gathers and hits among the `resident` most used lines
hipError_t l2_hits_device(const int64_t *rp_dev, const int32_t *ci_dev, const int64_t *r0_host, int nwin, int64_t W, int64_t ncols, bool f32, size_t resident,
                          double *refs, double *hits, hipStream_t st, Scratch lent = Scratch());

// ---- the chunk planner on the device (cvr_plan_dev.hip): the plan of plan_chunks from a device-resident row_ptr ----
struct Plan;
bool       plan_on_device_ok(int32_t S, int64_t max_rows = 0);
// scratch the caller may keep across calls: device bytes (grown on demand) and a pinned host buffer for the records coming back
struct PlanScratch { uint8_t *dev = nullptr; size_t dev_bytes = 0; uint8_t *pinned = nullptr; size_t pinned_bytes = 0;
                     bool borrowed = false; };      // borrowed: `dev` points INTO someone else's allocation -- never freed or grown here (a plan that needs more is an error)
void       free_plan_scratch(PlanScratch &ws);
// the planner's kernels enqueued without a synchronisation (cvr_fused.hip): the records stay on the device
struct DevicePlan {
    bool        declined = false;
    int64_t     bound = 0, thr = 0, max_rows = 0;       // room of the record arrays; the threshold and row cap in effect
    const void *chunks = nullptr;                       // [bound] cvr::Chunk records
    const void *shared = nullptr;                       // [bound] cvr::Shared records
    const unsigned long long *totals = nullptr;         // [4] chunks, cut rows, flags (1: host must plan, 2: tables too small), 2 * (most rows in a chunk) + (such a chunk has a pad segment)
    bool        chunks_shared_adjacent = false;
};
struct PlanTables {
    uint4 *desc = nullptr; uint2 *desc2 = nullptr; uint32_t *pad = nullptr; int64_t *nzb = nullptr; uint32_t room = 0; bool phased = false;
    unsigned long long *totals = nullptr;      // where the four totals go (null: the planner's scratch)
};
int64_t    plan_bound_device(int64_t nrows, int64_t nz_end, int32_t S, int64_t max_rows);       // room of the planner's record arrays / of the tables
size_t     plan_scratch_bytes(int64_t nrows, int64_t nz_end, int32_t S, int64_t max_rows);      // device scratch of one plan
hipError_t plan_chunks_device_enqueue(const int64_t *rp_dev, int64_t nrows, int64_t nz_end, int32_t S, int64_t thr, int64_t max_rows, hipStream_t st, PlanScratch *ws,
                                      DevicePlan *out, const PlanTables *tables);
hipError_t plan_chunks_device(const int64_t *rp_dev, int64_t nrows, int64_t nz_end, int32_t S, int64_t thr, int64_t max_rows, Plan *out, bool *fallback,
                              hipStream_t st, PlanScratch *ws = nullptr);
hipError_t max_row_device(const int64_t *rp_dev, int64_t nrows, int64_t *out, hipStream_t st);
hipError_t launch_shift_rows(const int64_t *src, int64_t n, int64_t base, int64_t *dst, hipStream_t st);
hipError_t launch_block_off(const uint32_t *rows, uint32_t n, uint32_t nblocks, uint32_t *out, hipStream_t st);

// ---- vector kernels of the iterative caller (cvr_iter.hip) ----
constexpr int kIterMaxParts = 64;
struct IterBounds { long long b[kIterMaxParts + 1]; };   // row offsets of the shards (by value into the kernel)
int        dot_partials();                                // doubles of scratch launch_dot needs
// out[0] = sum a[i] * b[i] (fp64 accumulation, fixed tree: bitwise reproducible); asynchronous
hipError_t launch_dot(const void *a, const void *b, int64_t n, bool f32, double *partial, double *out, hipStream_t st);
// one power-iteration step in one pass over x and y = A x: out[3 * kDotBlocks] = partial sums of x . y, y . y, x . x;
// x <- y / sqrt(sum of prev's y . y partials) (prev = the step before's `out`; null: x <- y)
int        power_partials();                              // doubles of one step's `out`
// (padded != null: y is the all-gathered vector of nparts slices of max_rows entries, read in row order through the bounds)
hipError_t launch_power_step(void *x, const void *y, int64_t n, bool f32, const double *prev, double *out, hipStream_t st, const IterBounds *padded = nullptr,
                             int nparts = 1, int64_t max_rows = 0);
// cells[0 .. 2] = the three sums of a step's partials
hipError_t launch_power_sums(const double *partial, double *cells, hipStream_t st);
// x[i] = y[i] / sqrt(norm2[0])
hipError_t launch_scale(void *x, const void *y, const double *norm2, int64_t n, bool f32, hipStream_t st);
// dense[bd.b[p] + i] = padded[p * max_rows + i], i < bd.b[p+1] - bd.b[p]
hipError_t launch_unpad(void *dense, const void *padded, const IterBounds &bd, int nparts, int64_t max_rows, bool f32, hipStream_t st);

// rows sorted by column? how many non-zeros in the fullest two adjacent bins of `half` columns of every 256 rows?  out2[2 * kProbeBlocks] = {unsorted flag, count} per workgroup
constexpr uint32_t kProbeBlocks = 1024;
hipError_t launch_probe(const int64_t *rp, const int32_t *ci, int64_t nrows, int64_t ncols, uint32_t half, unsigned long long *out2, hipStream_t st, bool out_zeroed = false);
// per chunk: its smallest column into cbase[k]; *wide (device u32, zeroed by the caller) gets 1 if any chunk spans 32 767 columns or more
hipError_t launch_chunk_span(const DeviceImage &img, const DeviceCsr &csr, uint32_t *cbase, uint32_t *wide, hipStream_t st);
// min / max of col_idx[n0 .. n1) into minmax[0..1] (device; initialised by the caller to INT_MAX / INT_MIN)
hipError_t launch_rows_check(const int64_t *rp, int64_t nrows, long long *out3, hipStream_t st);      // {rp[0], rp[nrows], first decreasing row or -1}
hipError_t launch_col_range(const int32_t *ci, int64_t n0, int64_t n1, int32_t *minmax, hipStream_t st);
hipError_t launch_dict_scan(const void *vals, int64_t n0, int64_t n1, bool f32, unsigned long long *table, uint32_t *flags, hipStream_t st);

// picks, per workgroup of img.wpb chunks, the window of img.win_elems consecutive columns that holds most
// of its non-zeros (LDS histogram over coarse column bins); writes img.win_base
hipError_t launch_window(const DeviceImage &img, const DeviceCsr &csr, hipStream_t st, const uint32_t *nchunks_dev = nullptr);

// column panels, one panel per XCD at a time: what differs between the eight panels of one launch (device array of 8; nchunks = 0: none)
struct PanelArgs { const uint8_t *stream; const uint4 *desc; const uint8_t *target; void *yext; uint32_t nchunks, ystage; const uint2 *desc2; uint32_t col_base, pad_col; const uint32_t *gbase; uint32_t gang0, pad_; };      // (gang0: the panel's first gang in the numbering of FuseArgs::range)      // (col_base, pad_col: interleaved panels keep panel-local columns)
// y_ext = A x  (+ the ordered fix-up of rows cut over chunks when img.nshared > 0 and with_fixup)
// multi != null: eight panels in one launch (plain layout, one chunk per workgroup, no LDS tables): workgroup b takes chunk b >> 3 of
// panel b & 7 of its round; multi[rounds][8]; multi_chunks = the most chunks any panel has (the rounds follow each other in ONE grid:
// no launch boundary between them); img = any of the panels (for what they share); y_ext and with_fixup unused
// The iterative caller's step inside the SpMV kernel's write-out (images with column phases, no rows cut over chunks, square matrix):
// per workgroup the partial sums of x . y, y . y, x . x over its rows -> out[set * nsets + workgroup], and x_next = y / ||y of the step
// before|| (prev: that step's partials, null = 1).  cvr_iter.hip's power_step_kernel is the same step as a pass of its own.
struct IterEpilogue { void *xnext = nullptr; const double *prev = nullptr; double *out = nullptr; uint32_t nsets = 0; };
bool iter_epilogue_ok(const DeviceImage &img);      // launch_spmv honours `epi` for this image
// The combine pass of column panels INSIDE the panel kernel (gang chunks; round 6): a gang that has stored its rows' partial sums counts itself in at
// every block of kCombineRows rows its sub-rows may lie in (range[gang] = {first, last block}; the ranges of a panel's gangs tile all blocks), and the
// workgroup whose count completes a block -- expect[block] = gangs of all panels that cover it -- adds that block's partial sums, panel by panel in panel
// order (the order of combine_kernel: the same bits), writes y and resets the counter.  Partial sums are stored and loaded past the non-coherent caches
// (sc1), as the hand-off between workgroups on different XCDs requires.  Tables made by launch_fuse_setup; y comes with the launch.
struct FuseArgs { uint32_t *cnt; const uint32_t *expect; const uint2 *range; const struct CombinePanel *panels; const uint32_t *block_off; uint32_t npanels, nblocks, nrows, ngangs; };
struct FusePanel { const uint4 *desc; const uint2 *desc2; const uint32_t *rows; uint32_t nchunks, gang0; };      // per panel: its chunk tables, the rows of its sub-rows
// range[] and expect[] of a handle's panels (gw chunks per gang); cnt zeroed.  Asynchronous on st.
hipError_t launch_fuse_setup(const FusePanel *panels_dev, uint32_t npanels, uint32_t gw, uint32_t ngangs, uint32_t nblocks, uint2 *range, uint32_t *first_last, uint32_t *expect, uint32_t *cnt, hipStream_t st);
// y[row] of the rows cut over chunks, once more after the fix-up launch has summed their carries (the fused combine read their partial sums too early):
// rows_list[i] = a global row; per panel the sub-row with that row (binary search), its partial sum added in panel order
hipError_t launch_fuse_cut_rows(const int64_t *shared, uint32_t nshared, const uint32_t *rows, uint32_t *out, hipStream_t st);      // out[i] = rows[shared[i].row]: the global rows of a panel's cut rows
hipError_t launch_fuse_patch(const uint32_t *rows_list, uint32_t nlist, const FusePanel *panels_dev, const struct CombinePanel *cpanels, const uint32_t *nsub, uint32_t npanels, void *y, bool f32, hipStream_t st);
hipError_t launch_spmv(const DeviceImage &img, const void *x_ext, void *y_ext, hipStream_t st, bool with_fixup = true, const PanelArgs *multi = nullptr, uint32_t multi_chunks = 0,
                       uint32_t multi_rounds = 1,
                       const IterEpilogue *epi = nullptr, const FuseArgs *fuse = nullptr, void *y_fused = nullptr);
size_t     spmv_lds_bytes(const DeviceImage &img);      // dynamic LDS of that launch

// column panels: one fix-up launch for all panels (each with its own y_ext inside the partial-sum buffer)
struct FixPart { const int64_t *shared; void *yext; uint32_t nshared, nrows; };
hipError_t launch_fixup_multi(const FixPart *parts, uint32_t nparts, uint32_t max_nshared, bool f32, hipStream_t st);

// column panels: y = sum over the panels, in panel order, of their partial sums.  Panel p's partial sum u stands at z[u]
// and belongs to row rows[u] (ascending); block_off[p * (nblocks + 1) + b] = first u of panel p with rows[u] >= b * kCombineRows.
constexpr int kCombineRows = 1024;
struct CombinePanel { const void *z; const uint16_t *rows; };          // rows: the LOW 16 BITS of the sub-rows' row numbers (a workgroup of the pass owns at most 8 192 consecutive rows: the difference to its first row, modulo 65 536, is the row's place there)
hipError_t launch_narrow_rows(const uint32_t *rows, size_t n, uint16_t *rows16, hipStream_t st);
// Rows cut over chunks, folded into the bitmap form of the pass (a handful per handle: the com-Orkut shape has two): entry = the panel, the row's place u among the
// panel's partial sums, its block of rows, the carries c0 .. c1 that make its sum (FixPart.shared) and where the panel's carries begin.  The workgroup of that
// block sums the carries with fixup_multi_kernel's own instructions and stores the result at the panel's place u before it loads its sums -- no fix-up launch in front of the pass.
struct CutEntry { uint32_t panel, u, block, carry_off; int64_t c0, c1; };
constexpr uint32_t kMaxCutFold = 8;
// out[0 .. total): the entries of all panels' cut rows (total <= kMaxCutFold, counted on the host); count: a zeroed device word.  rows32 / rows16_base: the
// 32-bit row numbers of all panels' partial sums and the array CombinePanel.rows points into (the same order).  Asynchronous on st.
hipError_t launch_cut_table(const struct FixPart *parts, uint32_t nparts, uint32_t max_nshared, const struct CombinePanel *panels, const uint16_t *rows16_base, const uint32_t *rows32, CutEntry *out, uint32_t *count, hipStream_t st);
hipError_t launch_combine(const CombinePanel *panels, uint32_t npanels, const uint32_t *block_off, void *y, uint32_t nrows, bool f32, hipStream_t st, int batch = 4, int mul = 1,
                          const uint32_t *bits = nullptr, const CutEntry *cut = nullptr, uint32_t ncut = 0);      // bits (mul = 1, <= 16 panels): the bitmap form -- a thread owns four rows, no row numbers read (combine_bits_kernel)
// bits[(p * nblocks + b) * 32 + w]: the rows b * kCombineRows + 32 w .. that have a partial sum in panel p (nblocks = ceil(nrows / kCombineRows)); asynchronous on st
hipError_t launch_combine_bits_build(const CombinePanel *panels, uint32_t npanels, const uint32_t *block_off, uint32_t nrows, uint32_t *bits, hipStream_t st);      // batch: panels whose loads share a round trip (4 or 8); mul: blocks of kCombineRows rows per workgroup (1 or 8)

// 16-B-per-lane streaming copy (roofline calibration)
hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t st);

void touch_convert_kernels();
void touch_plan_kernels();
void touch_spmv_kernels();

}  // namespace cvr

// cvr_internal.h -- what the translation units behind the C ABI share (cvr_capi / cvr_layout / cvr_panels / cvr_comm / cvr_tune /
// cvr_multi .hip): the handle, the per-image state, error helpers, and the declarations of each other's entry points.  Nothing
// here is part of include/cvr_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library itself is loaded on first use (rccl_api)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/cvr_amd.h"
#include "cvr_kernels.h"
#include "cvr_plan.h"

namespace cvrh {

extern thread_local char g_err[512];                 // cvr_last_error() of the calling thread
int fail(int code, const char *fmt, ...);           // formats g_err, returns code

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// roctx ranges around the phases of the path (the reference times them with microtime(), spmv.cpp:575/1009,
// 1033/1656), visible with `rocprofv3 --marker-trace`.  The roctx library is only loaded when CVR_ROCTX=1:
// linking it unconditionally costs every process seconds of profiler start-up.
struct Range {
    typedef int (*push_t)(const char *);
    typedef int (*pop_t)(void);
    static void resolve(push_t &push, pop_t &pop)
    {
        struct Fns { push_t p = nullptr; pop_t q = nullptr; };
        static const Fns f = [] {              // once, thread-safe
            Fns r;
            const char *e = getenv("CVR_ROCTX");
            if (e && atoi(e)) {
                void *lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
                if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (lib) { r.p = (push_t)dlsym(lib, "roctxRangePushA"); r.q = (pop_t)dlsym(lib, "roctxRangePop"); }
            }
            return r;
        }();
        push = f.p; pop = f.q;
    }
    pop_t pop_ = nullptr;
    explicit Range(const char *name)
    {
        push_t push;
        resolve(push, pop_);
        if (push && pop_) push(name); else pop_ = nullptr;
    }
    ~Range() { if (pop_) pop_(); }
};

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace cvrh

// cvr_options plus what cvr_create decides on the way and the profiling knobs it reads from the environment
struct IOpt : cvr_options {
    int32_t layout_auto_resident = 0;      // the automatic layout chose the "resident" form: every workgroup on a CU of its own at once
    int32_t debug_col_mask = 0;            // CVR_DEBUG_COL_MASK: folds the gather onto a 2^k-entry table (timing only: wrong y)
    int32_t panel_on_one_xcd = 0;          // this image is a column panel that will run on the workgroups of one XCD (run_spmv, d_multi)
    int32_t cus = 256, xcds = 8;           // the device's geometry (chip_of)
    int64_t col_span = 0;                  // > 0: the widest of the column panels that share a launch with this image: column and row fields are sized for it
};

// One CVR64 image on the device: the whole matrix, or one column panel of it (rows compacted to those that
// have a non-zero in the panel).
struct Part {
    cvr::DeviceImage img{};
    // device CSR + plan (dropped after conversion unless keep_csr)
    int64_t  *d_rp = nullptr;
    int32_t  *d_ci = nullptr;
    void     *d_va = nullptr;
    int64_t  *d_nzb = nullptr;
    uint32_t *d_pad = nullptr;
    size_t    stream_bytes = 0;
    int64_t   nrows = 0, nnz = 0, nnz_span = 0, nchunks = 0, nshared = 0, yext = 0;
    int64_t   zoff = 0;        // panels: where this part's y_ext starts in the partial-sum buffer z
    int32_t   multi_slot = -1; // panels that run one per XCD: round * 8 + XCD slot of this panel in cvr_handle::d_multi

    bool      csr_borrowed = false;      // d_ci / d_va point into the handle's split arena (column panels split on the device): not this part's to free
    bool      tables_borrowed = false;   // img.desc / desc2 / shared are slices of cvr_handle::panel_tables (they live as long as the handle)
    bool      rp_borrowed = false;       // d_rp, d_nzb, d_pad are slices of cvr_handle::panel_rp (the panels' row pointers in one allocation: a hipFree of a few megabytes takes ~0.2 ms, sixteen of them were 3 ms of cvr_preprocess)
    void release_csr()
    {
        if (d_rp && !rp_borrowed) (void)hipFree(d_rp);
        if (d_ci && !csr_borrowed) (void)hipFree(d_ci);
        if (d_va && !csr_borrowed) (void)hipFree(d_va);
        if (d_nzb && !rp_borrowed) (void)hipFree(d_nzb);          // (with the row pointers: slices of the same allocation)
        if (d_pad && !rp_borrowed) (void)hipFree(d_pad);
        rp_borrowed = false;
        d_rp = nullptr; d_ci = nullptr; d_va = nullptr; d_nzb = nullptr; d_pad = nullptr;
    }
    void release_all()
    {
        release_csr();
        if (img.stream) (void)hipFree(img.stream);
        if (img.desc && !tables_borrowed) (void)hipFree(img.desc);
        if (img.target) (void)hipFree(img.target);
        if (img.shared && !tables_borrowed) (void)hipFree(img.shared);
        if (img.win_base) (void)hipFree(img.win_base);
        if (img.desc2 && !tables_borrowed) (void)hipFree(img.desc2);
        tables_borrowed = false;
        if (img.cbase) (void)hipFree(img.cbase);
        if (img.hub_cols) (void)hipFree(img.hub_cols);
        if (img.hub_index) (void)hipFree(img.hub_index);
        if (img.hub_bitmap) (void)hipFree(img.hub_bitmap);
        if (img.hub_x) (void)hipFree(img.hub_x);
        if (img.prof) (void)hipFree(img.prof);
        if (img.gbase) (void)hipFree(img.gbase);
        img = cvr::DeviceImage{};
    }
};

struct cvr_handle {
    int               device = 0;
    hipStream_t       stream = nullptr;
    bool              converted = false;
    cvr_info          info{};
    std::vector<Part> parts;            // 1 part, or one per column panel
    // column panels: partial sums z (the panels' y_ext buffers, concatenated), per panel the rows of its sub-rows, and
    // where each block of kCombineRows rows starts in every panel (combine_kernel)
    void     *d_z = nullptr;
    uint32_t *d_rows = nullptr, *d_block_off = nullptr;
    uint32_t *d_cbits = nullptr;           // the rows that have a partial sum, a bit per (row, panel): the combine pass's bitmap form (setup_combine_bits; null: row numbers)
    cvr::CutEntry *d_cut = nullptr;        // ... and the few rows cut over chunks that pass folds in (ncut_fold of them; 0: the fix-up launch runs in front of the pass)
    uint32_t  ncut_fold = 0;
    uint16_t *d_rows16 = nullptr;          // the low halves of d_rows: what the combine pass reads (2 instead of 4 bytes per (row, panel) pair)
    int32_t  *split_ci = nullptr;          // the device split's column indices and values, panel after panel: the parts' CSR arrays are slices of these
    void     *split_va = nullptr;          // (freed with the parts' CSR: after the conversion, or with the handle when it keeps its CSR)
    void     *panel_tables = nullptr;      // the panels' chunk tables (desc, desc2, cut rows), one allocation (Part::tables_borrowed); freed with the handle
    int64_t  *panel_rp = nullptr;          // the panels' rebased row pointers, chunk starts and pad counts, one allocation (Part::rp_borrowed)
    void      release_split() { if (split_ci) (void)hipFree(split_ci); if (split_va) (void)hipFree(split_va); if (panel_rp) (void)hipFree(panel_rp); split_ci = nullptr; split_va = nullptr; panel_rp = nullptr; }
    cvr::CombinePanel *d_cpanels = nullptr;
    void     *d_dict = nullptr;           // value dictionary (sorted by bit pattern) shared by all parts, or null
    uint32_t  ndict = 0;
    cvr::FixPart *d_fixparts = nullptr;   // panels: the fix-up of every panel in one launch
    // panels, one per XCD at a time (cvr_kernels.h: PanelArgs): rounds of eight panels per launch; d_multi[round][8]
    cvr::PanelArgs       *d_multi = nullptr;
    // the combine pass inside the panel kernel (gang chunks; cvr_kernels.h: FuseArgs): tables in one allocation; null = the combine pass is a launch of its own
    void                 *fuse_mem = nullptr;
    cvr::FuseArgs        *d_fuse = nullptr;
    cvr::FusePanel       *d_fuse_panels = nullptr;
    uint32_t             *d_fuse_cut = nullptr, *d_fuse_nsub = nullptr;
    uint32_t              fuse_ncut = 0;
    std::vector<uint32_t> multi_chunks;   // per round: the most chunks any of its panels has
    uint32_t              multi_ystage = 0;
    uint32_t  max_nshared = 0;
    uint32_t *d_err = nullptr;
    void     *d_x = nullptr;            // x_ext: ncols + 1
    void     *d_y = nullptr;            // y_ext (1 part) or y (panels)
    size_t    vsz = 8;
    std::vector<hipEvent_t> events;
    hipEvent_t z_free = nullptr;         // column panels: recorded after the combine pass; the next SpMV (on any stream) waits for it
    bool       z_used = false;
    hipStream_t z_stream = nullptr;      // ... the stream of the launches so far: the event is recorded there when a launch comes on another one (run_spmv)
    int        combine_mul = 1;          // blocks of kCombineRows rows per workgroup of the combine pass: 8 when the panels' partial sums are fewer than the rows (mostly empty rows)
    int        combine_batch = 4;        // panels whose loads share a round trip in the combine pass (CVR_DEBUG=combine_batch=8: experiment)
    uint32_t   spmv_calls = 0;           // SpMVs launched so far (interleaved panels with ilv_flip: every other one walks the workgroups backwards)
    cvr::PlanScratch plan_ws;            // cvr_create only: scratch of the device planner (released before cvr_create returns)
    // cvr_create only: 32 KiB of device scratch for the small tables of its analysis passes (layout probe 16 KiB, dictionary
    // table 8 KiB + flags) and the host copies of the dictionary scan when it ran together with the probe
    uint8_t                        *d_small = nullptr;
    bool                            dict_scanned = false, small_clean = false;
    std::vector<unsigned long long> dict_tab;
    uint32_t                        dict_flags[2] = {0, 0};

    void  *seg_arena = nullptr;          // column phases: the conversion-time segment table, allocated by cvr_create (with the other buffers), released by cvr_preprocess
    size_t seg_arena_bytes = 0;
    bool   preconverted = false;         // cvr_create converted the image already (cvr_fused.hip): the first cvr_preprocess has nothing left to do
    IOpt opt_used;                       // the options the handle was created with (the image cache keys on them: cvr_image_io.hip)

    bool paneled() const { return parts.size() > 1; }
};

namespace cvrh {

// ---- cvr_capi.hip
// Geometry of a device as the layout rules need it: CUs from hipDeviceProp_t::multiProcessorCount; XCDs = CUs / 32 (gfx950 has 32
// CUs per XCD; the runtime reports no XCD count of its own), so a partitioned device (CPX: one XCD of 32 CUs) gets the arithmetic
// of its own size.  The XCD-aware parts (contiguous chunk ranges per XCD, one column panel per XCD) assume the eight XCDs of the
// whole chip and switch themselves off otherwise.  Cached per device; {256, 8} if the query fails.
struct Chip { int cus = 256, xcds = 8; };
Chip       chip_of(int device);
hipError_t run_spmv(cvr_handle *h, const void *x, void *y, hipStream_t st);
int        setup_fuse(cvr_handle *h);
// The combine pass's bitmap (cvr_kernels.h: launch_combine_bits_build) for a panelled handle with at most 16 panels and half or more of its (row, panel) pairs filled:
// allocated and filled on the handle's stream behind the combine tables.  nsub = the handle's partial sums (all panels).  CVR_DEBUG=combine_bits=0|1 overrides the rule.
int        setup_combine_bits(cvr_handle *h, int64_t nsub);                    // the fused combine's tables, for handles whose panels all carry gang chunks and run one per XCD (after d_multi)
void       ilv_runtime_settings(cvr_handle *h);          // helper wavefronts / sweep direction of interleaved images (launch parameters)      // y_ext = A x for the whole handle on `st`
IOpt       make_iopt(const cvr_options *in);
int        check_csr(const cvr_csr_view *c, bool columns_on_host = true);
int        check_columns_device(const int32_t *ci_dev, int64_t j0, int64_t j1, int64_t ncols);

// ---- cvr_layout.hip: the layout of one image (chunk length, workgroup shape, LDS budget, column phases, hub tables) and its device side
// the host side of one image: the chunk plan (from the host planner, or fetched from the device planner) and the per-chunk
// tables derived from it.  With a host row_ptr there is no device call and no error text: the panels of a host split plan
// their images on parallel threads.
struct PartPlan {
    int                   S = 0;
    cvr::Plan             plan;
    std::vector<uint32_t> desc, pad, desc2;
    std::vector<int64_t>  nzb;
    int64_t               max_nseg = 0, yext = 0;
    bool                  too_large = false;
    // LDS of the SpMV workgroup (160 KiB per CU)
    int      wpb = 1, phases = 1;
    int64_t  win = 0;              // x window, values
    int64_t  stage = 64;           // row sums (column phases: row accumulators) per wavefront
    int      col_bits = 31;        // column phases: bits of a column index (the row field of a segment's last column word starts there)
    bool     tag16 = false;        // column phases: the rows of the pieces in 16-bit tags of their own
    bool     lds_short = false;    // column phases do not fit beside the window
    bool     ilv = false;          // interleaved chunks (cvr_options.interleave): planned like an image with column phases (row cap, accumulators in LDS)
    bool     gang = false;         // gang chunks (cvr_options.gang): the workgroup's chunks are sorted together; the row field is the tag's share of a chunk
    int      plan_threads = 0;     // 0: the planner's own small team; 1: the caller plans several images side by side
    int64_t  hub_n = 0;            // hub table entries staged in LDS in front of the window (decided before planning)
    // the plan stayed on the device (plan_panels_batched): nzb / pad / desc / cut rows are in the part's buffers already, only the counts came back
    bool     tables_on_device = false;
    int64_t  dev_nchunks = 0, dev_nshared = 0;
};

// rows in device memory (rp == nullptr): row_ptr at dr->rp, first and last entry dr->nz0, dr->nz1; planned on the device
struct DevRows { const int64_t *rp = nullptr; int64_t nz0 = 0, nz1 = 0; hipStream_t st = nullptr; cvr::PlanScratch *ws = nullptr; };
// matrices of at least this many rows whose row_ptr is on the device anyway are planned there (cvr_plan_dev.hip)
constexpr int64_t kDevicePlanRows = 200000;
// (CVR_DEVICE_PLAN_ROWS overrides it: the fuzz tests send their small matrices through the device planner with 0)
inline int64_t device_plan_rows() { const char *e = cvr::debug_env("device_plan_rows"); return e ? atoll(e) : kDevicePlanRows; }

constexpr size_t kSmallProbe = 0, kSmallDictTab = 16 << 10, kSmallDictFlags = 24 << 10, kSmallBytes = 32 << 10;
constexpr size_t kPinnedProbe = 0, kPinnedDictTab = 16 << 10, kPinnedDictFlags = 24 << 10, kPinnedSmall = 32 << 10;      // in front of the planner's part of the pinned buffer
int        pick_steps(int64_t nslots_est, int64_t max_row = 0, double cus = 256.0, int64_t nrows = 0);
int        interleave_steps(int64_t nnz, int64_t nrows, bool f32, const IOpt &opt);
int        interleave_steps_panels(const std::vector<int64_t> &nnz, const std::vector<int64_t> &nsub, int64_t col_span, int rounds, bool f32, const IOpt &opt, int *generations = nullptr);
int        panel_generations(const std::vector<int64_t> &chunks, int rounds, int wpb, int cus_per_xcd, double *fullest = nullptr);
void       release_panel_plans(cvr_handle *h);
hipError_t plan_part(PartPlan &pp, int64_t nrows, int64_t ncols, bool f32, const int64_t *rp, const IOpt &opt, const DevRows *dr = nullptr);
int64_t    plan_layout(PartPlan &pp, int64_t ncols, bool f32, const IOpt &opt);
void       plan_stage(PartPlan &pp, bool f32);
bool       resident_candidate(double slots, int cus, int *best_w, int *best_S);
bool       resident_out_of_reach(int64_t nrows, int64_t nnz, int64_t ncols, bool f32, const IOpt &opt);
constexpr double kNoWindowPanelBytes = 8e6;  // ... and from here on for matrices whose non-zeros are not near the diagonal (no use for the resident layout's window: cvr_create)
constexpr double kMidPanelBytes = 12e6;      // x from here to 24 MB: the panel rule runs only for matrices beyond the resident layout
int        setup_image(cvr_handle *h, Part &part, const PartPlan &pp, int64_t nrows, int64_t ncols, bool f32, int64_t nchunks, int64_t nshared, const IOpt &opt, const IOpt &popt);
// cvr_fused.hip: analysis, plan and conversion of a single resident-layout image as one submission (no host round trip between planner
// and converter); *taken = false: the matrix is not of that kind (or the layout was not confirmed) and the staged path takes over
int        build_part_fused(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, const IOpt &opt, IOpt &popt, bool *taken);
hipStream_t side_stream(int device, int which = 0);      // two streams per device beside the handles' own (analysis passes side by side)
hipError_t acquire_stream(int device, hipStream_t *out);
void       release_stream(int device, hipStream_t s);
hipError_t enqueue_dict_scan(cvr_handle *h, const void *d_va, int64_t nz0, int64_t nz1, bool f32, bool first, unsigned long long *tab_host, uint32_t *flags_host, bool last,
                             hipStream_t st);
int        auto_layout(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, IOpt &opt,
                       const std::function<int(const IOpt &)> *meanwhile = nullptr);
int        choose_hubs(cvr_handle *h, Part &part, const int32_t *d_ci, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, IOpt &opt, PartPlan &pp,
                       bool allow_reorder);
int        plan_panels_batched(cvr_handle *h, const cvr::DeviceSplit &d, const std::vector<int64_t> &nsubs, const std::vector<int64_t> &pcols, bool f32, const std::vector<IOpt> &popts,
                               std::vector<PartPlan> &pps, std::vector<DevRows> &drs, bool *done);
int        build_part(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *ci, const void *va,
                      hipMemcpyKind civa_kind, bool f32, const IOpt &opt, double *plan_s, PartPlan *planned = nullptr, const DevRows *dr = nullptr);
int        finish_part(cvr_handle *h, Part &part);

// ---- cvr_panels.hip: column panels from host arrays, the panel rule
// host arrays without value-initialisation (hundreds of MB: a zero-fill pass per array is measurable)
template <typename T> struct Raw {
    std::unique_ptr<T[]> p;
    size_t               n = 0;
    void     alloc(size_t m) { p.reset(new T[m ? m : 1]); n = m; }
    T       *data() { return p.get(); }
    const T *data() const { return p.get(); }
    size_t   size() const { return n; }
    T       &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

struct PanelSplit {
    std::vector<Raw<int64_t>>  rp;
    std::vector<Raw<int32_t>>  ci;
    std::vector<Raw<uint64_t>> va;     // values as raw 8-byte words (fp32: two per word)
    std::vector<Raw<uint32_t>> rows;   // compact sub-row -> row
};

void       split_panels(const cvr_csr_view &v, int P, PanelSplit &out);
double     l2_miss_estimate(const cvr_csr_view &v);
hipError_t l2_miss_estimate_dev(const int64_t *rp_dev, const int32_t *ci_dev, int64_t nrows, int64_t ncols, bool f32, hipStream_t st, double *miss, cvr::Scratch lent = cvr::Scratch());
int        panels_from_miss(double xb, double miss);
double     pairs_per_nnz(const cvr_csr_view &v, int64_t width);
hipError_t pairs_per_nnz_dev(const int64_t *rp_dev, const int32_t *ci_dev, int64_t nrows, int64_t width, hipStream_t st, double *out, cvr::Scratch lent = cvr::Scratch());
bool       panels_pay(double miss, double pairs_per_nonzero);
bool       thin_lists(int P, double xbytes, int64_t nnz, int cus, double pairs_per_nonzero);      // gang chunks with fewer than two non-zeros per line of a panel's slice, few partial sums: panels half as wide (cvr_panels.hip)
int        auto_panels(const cvr_csr_view &v, double *miss_out);
int        xcd_panel_count(int P, double xbytes);

// ---- cvr_comm.hip: RCCL, loaded on first use
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId)    get_unique_id = nullptr;
    decltype(&ncclCommInitRank)   comm_init_rank = nullptr;
    decltype(&ncclCommDestroy)    comm_destroy = nullptr;
    decltype(&ncclAllGather)      all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommInitAll)    comm_init_all = nullptr;      // one process driving several GPUs (cvr_multi.hip)
    decltype(&ncclGroupStart)     group_start = nullptr;
    decltype(&ncclGroupEnd)       group_end = nullptr;
    // optional (cvr_comm_info: what the library itself reports about a communicator)
    decltype(&ncclCommCount)      comm_count = nullptr;
    decltype(&ncclCommUserRank)   comm_user_rank = nullptr;
    decltype(&ncclGetVersion)     get_version = nullptr;
};
const RcclApi *rccl_api();      // null when librccl cannot be loaded

#define RCCL_TRY(api, expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, (api)->error_string(r_), __FILE__, __LINE__); \
    } while (0)


}  // namespace cvrh

struct cvr_comm {
    ncclComm_t  comm = nullptr;
    int         nranks = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;                 // the collectives of cvr_spmv_gather_repeat run here
    hipEvent_t  ready[2] = {nullptr, nullptr};    // y_dev[b] computed
    hipEvent_t  done[2] = {nullptr, nullptr};     // gather of y_dev[b] into yall_dev[b] finished
    bool        pending[2] = {false, false};
};

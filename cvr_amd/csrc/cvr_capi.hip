// cvr_capi.hip -- the C ABI of include/cvr_amd.h: handle life cycle, device memory, timing.
// Host orchestration that the reference keeps in main() (allocation block spmv.cpp:1777-1829, calls at
// spmv.cpp:1857 and 1882) lives behind the handle here; the caller keeps only CSR, x and y.
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
#include <thread>
#include <vector>

#include "../../include/cvr_amd.h"
#include "cvr_kernels.h"
#include "cvr_plan.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

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

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// cvr_options plus what cvr_create decides on the way and the profiling knobs it reads from the environment
struct IOpt : cvr_options {
    int32_t layout_auto_resident = 0;      // the automatic layout chose the "resident" form: every workgroup on a CU of its own at once
    int32_t stream_ahead = 0;              // CVR_DEBUG_STREAM_AHEAD: groups the matrix stream runs ahead of the x gather: 0 / 1 = one, >= 2 = three
    int32_t gather_depth = 0;              // CVR_DEBUG_GATHER_DEPTH: groups the x gather runs ahead of the FMAs: 1 or 2
    int32_t debug_col_mask = 0;            // CVR_DEBUG_COL_MASK: folds the gather onto a 2^k-entry table (timing only: wrong y)
    int32_t panel_on_one_xcd = 0;          // this image is a column panel that will run on the workgroups of one XCD (run_spmv, d_multi)
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

    void release_csr()
    {
        if (d_rp) (void)hipFree(d_rp);
        if (d_ci) (void)hipFree(d_ci);
        if (d_va) (void)hipFree(d_va);
        if (d_nzb) (void)hipFree(d_nzb);
        if (d_pad) (void)hipFree(d_pad);
        d_rp = nullptr; d_ci = nullptr; d_va = nullptr; d_nzb = nullptr; d_pad = nullptr;
    }
    void release_all()
    {
        release_csr();
        if (img.stream) (void)hipFree(img.stream);
        if (img.desc) (void)hipFree(img.desc);
        if (img.target) (void)hipFree(img.target);
        if (img.shared) (void)hipFree(img.shared);
        if (img.win_base) (void)hipFree(img.win_base);
        if (img.desc2) (void)hipFree(img.desc2);
        if (img.pace) (void)hipFree(img.pace);
        delete img.pace_epoch;
        if (img.cbase) (void)hipFree(img.cbase);
        if (img.hub_cols) (void)hipFree(img.hub_cols);
        if (img.hub_index) (void)hipFree(img.hub_index);
        if (img.hub_bitmap) (void)hipFree(img.hub_bitmap);
        if (img.hub_x) (void)hipFree(img.hub_x);
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
    cvr::CombinePanel *d_cpanels = nullptr;
    void     *d_dict = nullptr;           // value dictionary (sorted by bit pattern) shared by all parts, or null
    uint32_t  ndict = 0;
    cvr::FixPart *d_fixparts = nullptr;   // panels: the fix-up of every panel in one launch
    // panels, one per XCD at a time (cvr_kernels.h: PanelArgs): rounds of eight panels per launch; d_multi[round][8]
    cvr::PanelArgs       *d_multi = nullptr;
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
    cvr::PlanScratch plan_ws;            // cvr_create only: scratch of the device planner (released before cvr_create returns)
    // cvr_create only: 32 KiB of device scratch for the small tables of its analysis passes (layout probe 16 KiB, dictionary
    // table 8 KiB + flags) and the host copies of the dictionary scan when it ran together with the probe
    uint8_t                        *d_small = nullptr;
    bool                            dict_scanned = false, small_clean = false;
    std::vector<unsigned long long> dict_tab;
    uint32_t                        dict_flags[2] = {0, 0};

    bool paneled() const { return parts.size() > 1; }
};

namespace {

// y_ext = A x for the whole handle on `st`: one SpMV launch, or one per column panel followed by the combine
hipError_t run_spmv(cvr_handle *h, const void *x, void *y, hipStream_t st)
{
    if (!h->paneled()) {
        if (h->parts.empty()) return hipSuccess;
        if (h->parts[0].img.hub_n == 0) return cvr::launch_spmv(h->parts[0].img, x, y, st);
        // the hub table's compacted copy of x is a buffer of the handle: launches on different streams are kept apart
        if (h->z_used) { const hipError_t e = hipStreamWaitEvent(st, h->z_free, 0); if (e != hipSuccess) return e; }
        const hipError_t e = cvr::launch_spmv(h->parts[0].img, x, y, st);
        if (e != hipSuccess) return e;
        h->z_used = true;
        return hipEventRecord(h->z_free, st);
    }
    // the panels' partial sums share one buffer (h->d_z): a launch on another stream must not start before the combine pass of
    // the previous one has read them
    if (h->z_used) { const hipError_t e = hipStreamWaitEvent(st, h->z_free, 0); if (e != hipSuccess) return e; }
    if (h->d_multi) {          // eight panels per launch, panel b & 7 on the XCD of the workgroups b
        cvr::DeviceImage shared = h->parts[0].img;
        shared.ystage = h->multi_ystage;
        for (size_t r = 0; r < h->multi_chunks.size(); r++) {
            if (h->multi_chunks[r] == 0) continue;
            const hipError_t e = cvr::launch_spmv(shared, x, nullptr, st, false, h->d_multi + 8 * r, h->multi_chunks[r]);
            if (e != hipSuccess) return e;
        }
    } else
        for (const Part &p : h->parts) {
            hipError_t e = cvr::launch_spmv(p.img, x, static_cast<uint8_t *>(h->d_z) + (size_t)p.zoff * h->vsz, st, false);
            if (e != hipSuccess) return e;
        }
    hipError_t e = cvr::launch_fixup_multi(h->d_fixparts, (uint32_t)h->parts.size(), h->max_nshared, h->vsz == 4, st);
    if (e != hipSuccess) return e;
    e = cvr::launch_combine(h->d_cpanels, (uint32_t)h->parts.size(), h->d_block_off, y, (uint32_t)h->info.nrows, h->vsz == 4, st);
    if (e != hipSuccess) return e;
    h->z_used = true;
    return hipEventRecord(h->z_free, st);
}

}  // namespace

extern "C" {

const char *cvr_last_error(void) { return g_err; }
const char *cvr_version(void) { return "cvr_amd 0.1 (gfx950, CVR64)"; }

void cvr_default_options(cvr_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->device = 0;
    o->steps_per_chunk = 0;
    o->split_threshold = 0;
    o->xcd_swizzle = -1;
    o->x_window = -1;
    o->col_panels = -1;
    o->value_dict = -1;
    o->col_phases = -1;
    o->hub_table = -1;
    o->narrow_cols = -1;
    o->hub_reorder = -1;
    o->row_tags16 = -1;
    o->row_bands = -1;
    o->piece_max = -1;
}

}  // extern "C"

static IOpt make_iopt(const cvr_options *in)
{
    IOpt o;
    if (in) static_cast<cvr_options &>(o) = *in; else cvr_default_options(&o);
    auto env = [](const char *name) { const char *e = getenv(name); return e ? (int32_t)strtol(e, nullptr, 0) : 0; };
    o.stream_ahead = env("CVR_DEBUG_STREAM_AHEAD");
    o.gather_depth = env("CVR_DEBUG_GATHER_DEPTH");
    o.debug_col_mask = env("CVR_DEBUG_COL_MASK");
    return o;
}

extern "C" {

int cvr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int64_t cvr_plan_bound(int64_t nrows, int64_t nnz, int32_t S) { return cvr::plan_bound(nrows, nnz, S); }

static int check_csr(const cvr_csr_view *c, bool columns_on_host = true)
{
    if (!c || c->nrows < 0 || c->ncols < 0) return fail(CVR_ERR_INVALID, "null or negative-size CSR view");
    if (c->nrows > 0 && !c->row_ptr) return fail(CVR_ERR_INVALID, "row_ptr is null");
    if (c->ncols >= (int64_t)0x7fffffff) return fail(CVR_ERR_INVALID, "ncols must be < 2^31 - 1 (bit 31 of a column word is the segment-end flag)");
    if ((uint64_t)(c->ncols + 1) * (c->is_f32 ? 4u : 8u) > 0xffffffffull)
        return fail(CVR_ERR_INVALID, "x (%lld values) exceeds the 4 GiB a buffer descriptor addresses; shard the columns", (long long)c->ncols);
    if (c->nrows == 0) return CVR_OK;
    if (c->row_ptr[0] < 0) return fail(CVR_ERR_INVALID, "row_ptr[0] < 0");
    for (int64_t r = 0; r < c->nrows; r++)
        if (c->row_ptr[r + 1] < c->row_ptr[r]) return fail(CVR_ERR_INVALID, "row_ptr decreases at row %lld", (long long)r);
    const int64_t nnz = c->row_ptr[c->nrows];
    if (nnz > 0 && (!c->col_idx || !c->vals)) return fail(CVR_ERR_INVALID, "col_idx / vals is null");
    if (!columns_on_host) return CVR_OK;       // device arrays: the range check is a kernel (check_columns_device)
    // the first offending position, searched by a few threads on large matrices (69 M columns: 17 ms on one core)
    const int64_t j0 = c->row_ptr[0];
    int           T = (int)std::thread::hardware_concurrency();
    if (T > 16) T = 16;
    if (T < 1 || nnz - j0 < (1 << 22)) T = 1;
    std::vector<int64_t> bad((size_t)T, -1);
    auto scan = [&](int t) {
        const int64_t a = j0 + (nnz - j0) * t / T, b = j0 + (nnz - j0) * (t + 1) / T;
        const int32_t nc = (int32_t)c->ncols;
        for (int64_t j = a; j < b; j++)
            if ((uint32_t)c->col_idx[j] >= (uint32_t)nc) { bad[(size_t)t] = j; return; }     // negative or >= ncols
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(scan, t);
        scan(0);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; t++)
        if (bad[(size_t)t] >= 0)
            return fail(CVR_ERR_INVALID, "col_idx[%lld] = %d outside [0, %lld)", (long long)bad[(size_t)t], c->col_idx[bad[(size_t)t]], (long long)c->ncols);
    return CVR_OK;
}

// the column range check for col_idx in device memory
static int check_columns_device(const int32_t *ci_dev, int64_t j0, int64_t j1, int64_t ncols)
{
    if (j1 <= j0) return CVR_OK;
    int32_t *mm = nullptr;
    int32_t  host[2] = {0x7fffffff, (int32_t)0x80000000};
    HIP_TRY(hipMalloc(&mm, sizeof(host)));
    hipError_t e = hipMemcpy(mm, host, sizeof(host), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = cvr::launch_col_range(ci_dev, j0, j1, mm, nullptr);
    if (e == hipSuccess) e = hipMemcpy(host, mm, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipFree(mm);
    if (e != hipSuccess) return fail(CVR_ERR_HIP, "column range check: %s", hipGetErrorString(e));
    if (host[0] < 0 || host[1] >= ncols) return fail(CVR_ERR_INVALID, "col_idx holds %d .. %d, outside [0, %lld)", host[0], host[1], (long long)ncols);
    return CVR_OK;
}

int64_t cvr_plan_chunks(int64_t nrows, const int64_t *row_ptr, int32_t S, int64_t thr, int64_t *nz_begin,
                        int64_t *row_first, int64_t *nseg, int64_t *pad_cnt)
{
    if (nrows < 0 || (nrows > 0 && !row_ptr) || S < 4 || S % 4) return fail(CVR_ERR_INVALID, "bad planner arguments");
    const cvr::Plan p = cvr::plan_chunks(nrows, row_ptr, S, thr);
    const int64_t   n = (int64_t)p.chunks.size();
    for (int64_t k = 0; k < n; k++) {
        if (nz_begin) nz_begin[k] = p.chunks[k].nz_begin;
        if (row_first) row_first[k] = p.chunks[k].row_first;
        if (nseg) nseg[k] = p.chunks[k].nseg;
        if (pad_cnt) pad_cnt[k] = p.chunks[k].pad_cnt;
    }
    if (nz_begin) nz_begin[n] = p.nz_end;
    return n;
}

// diagnostics: the plan of the device planner (cvr_plan_dev.hip) against the host planner's for the same row_ptr, field by
// field; seconds of both (the device figure includes the two synchronisations, not the upload of row_ptr)
int cvr_plan_selfcheck(int device, int64_t nrows, const int64_t *row_ptr, int32_t S, int64_t thr, int64_t max_rows, double *host_s, double *device_s,
                       int64_t *nchunks)
{
    if (nrows < 0 || (nrows > 0 && !row_ptr) || S < 4 || S % 4) return fail(CVR_ERR_INVALID, "bad planner arguments");
    if (device < 0 || device >= cvr_device_count()) return fail(CVR_ERR_NO_DEVICE, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    const double    t0 = now_s();
    const cvr::Plan ph = cvr::plan_chunks(nrows, row_ptr, S, thr, max_rows);
    const double    t1 = now_s();
    if (host_s) *host_s = t1 - t0;
    if (nchunks) *nchunks = (int64_t)ph.chunks.size();
    int64_t *d_rp = nullptr;
    HIP_TRY(hipMalloc(&d_rp, sizeof(int64_t) * ((size_t)nrows + 1)));
    hipError_t e = nrows > 0 ? hipMemcpy(d_rp, row_ptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice) : hipSuccess;
    cvr::Plan        pd;
    bool             fallback = false;
    double           best = 1e30;
    cvr::PlanScratch ws;             // as cvr_create keeps it: device scratch re-used, records through a pinned buffer
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ws.pinned), ws.pinned_bytes = (size_t)640 << 10, hipHostMallocDefault);
    for (int rep = 0; rep < 3 && e == hipSuccess; rep++) {
        const double t2 = now_s();
        e = cvr::plan_chunks_device(d_rp, nrows, nrows ? row_ptr[nrows] : 0, S, thr, max_rows, &pd, &fallback, nullptr, rep ? &ws : nullptr);
        best = std::min(best, now_s() - t2);
    }
    cvr::free_plan_scratch(ws);
    (void)hipFree(d_rp);
    if (e != hipSuccess) return fail(CVR_ERR_HIP, "device planner: %s", hipGetErrorString(e));
    if (device_s) *device_s = best;
    if (fallback) return fail(CVR_ERR_STATE, "device planner declined (row block beyond 32-bit slot positions, or S too large)");
    if (pd.S != ph.S || pd.thr != ph.thr || pd.max_rows != ph.max_rows || pd.nz_end != ph.nz_end) return fail(CVR_ERR_INTERNAL, "plan parameters differ");
    if (pd.chunks.size() != ph.chunks.size()) return fail(CVR_ERR_INTERNAL, "device plan has %zu chunks, host plan %zu", pd.chunks.size(), ph.chunks.size());
    for (size_t k = 0; k < ph.chunks.size(); k++) {
        const cvr::Chunk &a = ph.chunks[k], &b = pd.chunks[k];
        if (a.nz_begin != b.nz_begin || a.row_first != b.row_first || a.nrows_in != b.nrows_in || a.nseg != b.nseg || a.pad_cnt != b.pad_cnt ||
            a.head_shared != b.head_shared || a.tail_shared != b.tail_shared)
            return fail(CVR_ERR_INTERNAL, "chunk %zu differs: host {nz %lld row %lld rows %lld seg %lld pad %lld %d%d} device {nz %lld row %lld rows %lld seg %lld pad %lld %d%d}", k,
                        (long long)a.nz_begin, (long long)a.row_first, (long long)a.nrows_in, (long long)a.nseg, (long long)a.pad_cnt, a.head_shared, a.tail_shared,
                        (long long)b.nz_begin, (long long)b.row_first, (long long)b.nrows_in, (long long)b.nseg, (long long)b.pad_cnt, b.head_shared, b.tail_shared);
    }
    if (pd.shared.size() != ph.shared.size()) return fail(CVR_ERR_INTERNAL, "device plan has %zu cut rows, host plan %zu", pd.shared.size(), ph.shared.size());
    for (size_t k = 0; k < ph.shared.size(); k++)
        if (pd.shared[k].row != ph.shared[k].row || pd.shared[k].c0 != ph.shared[k].c0 || pd.shared[k].c1 != ph.shared[k].c1)
            return fail(CVR_ERR_INTERNAL, "cut row %zu differs: host {%lld %lld %lld} device {%lld %lld %lld}", k, (long long)ph.shared[k].row, (long long)ph.shared[k].c0,
                        (long long)ph.shared[k].c1, (long long)pd.shared[k].row, (long long)pd.shared[k].c0, (long long)pd.shared[k].c1);
    return CVR_OK;
}

static int pick_steps(int64_t nslots_est, int64_t max_row = 0, double cus = 256.0)
{
    // The plain layout (one chunk per workgroup).  Images of more than 12 chunks per CU at S = 32 run in rounds and take
    // S = 32 (LiveJournal panels, R-MAT, banded: within 1 % of the best S, profiles/r02_steps_rule_check.log,
    // r01_steps_large_matrices.log).  Smaller ones are resident at once: what decides there is (1) that no row is cut over
    // chunks -- a cut row brings the fix-up kernel, a second launch worth 1.9 us on a 8-us SpMV -- so 16 S >= the longest
    // row, and (2) beyond that as many chunks as possible, i.e. the smallest such S (web-Google-shaped matrices of 0.6 M and
    // 1.3 M non-zeros: S = 28 is the best of 8 .. 64, 7.9 and 9.5 us; the round-1 fit on shards of one matrix took 24 and 44:
    // 9.8 and 11.8 us).  cvr_tune measures instead.
    const double kCus = cus;             // (a column panel that runs on one XCD counts its chunks against that XCD's 32 CUs)
    auto chunks = [&](int S) { return (double)nslots_est * 1.004 / (64.0 * S) + 1.0; };
    if (chunks(32) > kCus * 12.0) return 32;
    int S = (int)std::min<int64_t>(64, std::max<int64_t>(12, ((max_row + 15) / 16 + 3) / 4 * 4));
    while (S < 64 && chunks(S) > kCus * 12.0) S += 4;
    return S;
}

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
    int      plan_threads = 0;     // 0: the planner's own small team; 1: the caller plans several images side by side
    int64_t  hub_n = 0;            // hub table entries staged in LDS in front of the window (decided before planning)
};

// rows in device memory (rp == nullptr): row_ptr at dr->rp, first and last entry dr->nz0, dr->nz1; planned on the device
struct DevRows { const int64_t *rp = nullptr; int64_t nz0 = 0, nz1 = 0; hipStream_t st = nullptr; cvr::PlanScratch *ws = nullptr; };
// matrices of at least this many rows whose row_ptr is on the device anyway are planned there (cvr_plan_dev.hip)
constexpr int64_t kDevicePlanRows = 200000;
// (CVR_DEVICE_PLAN_ROWS overrides it: the fuzz tests send their small matrices through the device planner with 0)
static int64_t device_plan_rows() { const char *e = getenv("CVR_DEVICE_PLAN_ROWS"); return e ? atoll(e) : kDevicePlanRows; }

static hipError_t plan_part(PartPlan &pp, int64_t nrows, int64_t ncols, bool f32, const int64_t *rp, const IOpt &opt, const DevRows *dr = nullptr)
{
    const int64_t nz0 = rp ? (nrows ? rp[0] : 0) : dr->nz0, nz1 = rp ? (nrows ? rp[nrows] : 0) : dr->nz1;
    pp.S = opt.steps_per_chunk;
    if (pp.S == 0) {
        int64_t max_row = 0;
        const double cus = opt.panel_on_one_xcd ? 32.0 : 256.0;
        if ((double)(nz1 - nz0 + nrows / 4) / (64.0 * 32.0) <= cus * 12.0) {    // (only where the rule weighs single launches)
            if (rp) for (int64_t r = 0; r < nrows; r++) max_row = std::max(max_row, rp[r + 1] - rp[r]);
            else { const hipError_t e = cvr::max_row_device(dr->rp, nrows, &max_row, dr->st); if (e != hipSuccess) return e; }
        }
        pp.S = pick_steps(nz1 - nz0 + nrows / 4, max_row, cus);
    }
    // Wavefronts (consecutive chunks) per SpMV workgroup: 1 by default; more only pay together with an LDS window of x,
    // which the workgroup's chunks then share (profiles/r02_wg_window_sweep.log).
    pp.wpb = std::min(std::max(opt.waves_per_block, 1), cvr::kMaxWavesPerBlock);
    pp.phases = std::min(std::max(opt.col_phases, 1), 64);
    if (ncols < 64 * pp.phases) pp.phases = 1;
    if (pp.hub_n > 0) pp.phases = 1;               // (the hub flag and the row field of a phased image share bits of the column word)
    // LDS window of x per workgroup (off by default): `win` consecutive values of x staged with coalesced loads; gathers
    // inside it are served by ds_read instead of a 128-byte L1 fill each.
    pp.win = std::min<int64_t>(opt.x_window < 0 ? 0 : opt.x_window, ncols + 1) & ~(int64_t)3;      // whole 16-byte loads, inside x_ext
    const int64_t vs = f32 ? 4 : 8;
    int64_t       max_rows = 0;
    if (pp.phases > 1) {
        // column phases: every chunk accumulates its rows in LDS, so the planner caps the rows of a chunk at what is left of
        // the 160 KiB beside steal slots, dictionary and window -- and at what the row field of a segment's last column
        // word can hold (the bits between the column index and the end flag)
        pp.col_bits = 1;
        while (((int64_t)1 << pp.col_bits) <= ncols) pp.col_bits++;
        const int64_t row_field = pp.col_bits < 31 ? ((int64_t)1 << (31 - pp.col_bits)) - 1 : 0;
        auto rows_for = [&](int64_t win) {
            const int64_t left = (int64_t)cvr::kLdsBytes - (cvr::kDictMax + win + 4) * vs;      // (no steal slots: spmv_seg_kernel)
            return std::min<int64_t>((left / pp.wpb / vs) & ~(int64_t)3, cvr::kYStageMax);
        };
        while (pp.win > 0 && rows_for(pp.win) < 512) pp.win = (pp.win - 1024 > 0 ? pp.win - 1024 : 0) & ~(int64_t)3;      // the window gives way
        // a chunk of S steps holds at most 64 S rows: no need for more accumulators than that (keeps the LDS small)
        const int64_t want = std::min<int64_t>(rows_for(pp.win), ((int64_t)cvr::kLanes * pp.S + 1 + 3) & ~(int64_t)3);
        // wide row tags (16 bits of their own per slot) when the column word has no room for the rows such a chunk may hold
        pp.tag16 = opt.row_tags16 > 0 || (opt.row_tags16 < 0 && row_field + 1 < want);
        if (pp.tag16) pp.col_bits = 31;
        pp.stage = std::min<int64_t>(want, pp.tag16 ? (int64_t)65532 : (row_field + 1) & ~(int64_t)3);
        if (pp.stage < 64) { pp.lds_short = true; pp.phases = 1; pp.stage = 64; }
        else max_rows = pp.stage - 1;                 // + the dump entry of the pad segment
    }
    if (rp) {
        pp.plan = cvr::plan_chunks(nrows, rp, pp.S, opt.split_threshold, max_rows, pp.plan_threads);
    } else {
        bool             declined = false;
        const hipError_t e = cvr::plan_chunks_device(dr->rp, nrows, nz1, pp.S, opt.split_threshold, max_rows, &pp.plan, &declined, dr->st, dr->ws);
        if (e != hipSuccess) return e;
        if (declined) {       // (chunks beyond the 15-bit jump, or a row block beyond 32-bit slot positions): the row pointers come to the host after all
            std::vector<int64_t> hrp((size_t)nrows + 1);
            const hipError_t     e2 = hipMemcpy(hrp.data(), dr->rp, sizeof(int64_t) * hrp.size(), hipMemcpyDeviceToHost);
            if (e2 != hipSuccess) return e2;
            pp.plan = cvr::plan_chunks(nrows, hrp.data(), pp.S, opt.split_threshold, max_rows, pp.plan_threads);
        }
    }
    const cvr::Plan &plan = pp.plan;
    const int64_t    nchunks = (int64_t)plan.chunks.size();
    pp.yext = nrows + 1 + 2 * nchunks;
    if (pp.yext >= (int64_t)0xffffffffu || nchunks >= (int64_t)0x7fffffff) { pp.too_large = true; return hipSuccess; }
    pp.desc.resize((size_t)nchunks * 4);
    if (pp.phases > 1) pp.desc2.resize((size_t)nchunks * 2, 0u);
    pp.pad.resize((size_t)nchunks);
    pp.nzb.resize((size_t)nchunks + 1);
    for (int64_t k = 0; k < nchunks; k++) {
        const cvr::Chunk &c = plan.chunks[(size_t)k];
        pp.max_nseg = std::max(pp.max_nseg, c.nseg);
        pp.desc[4 * k + 0] = (uint32_t)c.row_first;
        pp.desc[4 * k + 1] = (uint32_t)c.nseg;
        // where segment q writes: a row begun earlier -> carry_head(k); a row continued later -> carry_tail(k);
        // the pad segment -> dump; else its row.  head_dest / last_dest are that rule at q = 0 and q = nseg-1.
        auto dest = [&](int64_t q) -> uint32_t {
            if (q >= c.nrows_in) return (uint32_t)nrows;
            if (q == 0 && c.head_shared) return (uint32_t)(nrows + 1 + 2 * k);
            if (q == c.nrows_in - 1 && c.tail_shared) return (uint32_t)(nrows + 1 + 2 * k + 1);
            return (uint32_t)(c.row_first + q);
        };
        pp.desc[4 * k + 2] = dest(0);
        pp.desc[4 * k + 3] = dest(c.nseg - 1);
        if (pp.phases > 1) {            // column phases: head / last_dest belong to the first / last ROW; desc.y and desc2.x come from the device
            pp.desc[4 * k + 3] = dest(c.nrows_in - 1);
            pp.desc2[2 * k + 1] = (uint32_t)c.nrows_in;
        }
        pp.pad[(size_t)k] = (uint32_t)c.pad_cnt;
        pp.nzb[(size_t)k] = c.nz_begin;
    }
    pp.nzb[(size_t)nchunks] = plan.nz_end;
    if (pp.phases > 1) {          // no more accumulators than the fullest chunk has rows (+ the dump entry): the cap stays what no chunk exceeds
        int64_t most = 0;
        for (const cvr::Chunk &c : plan.chunks) most = std::max(most, c.nrows_in);
        pp.stage = std::min<int64_t>(pp.stage, std::max<int64_t>(64, (most + 1 + 3) & ~(int64_t)3));
    }
    if (pp.phases == 1) {
        // LDS budget without phases: steal slots and dictionary are fixed; the row-sum stage is sized for the chunk with the
        // most segments, so every chunk writes its y coalesced (chunks of very short rows beyond the stage store directly);
        // the window takes what it asked for, the stage at least 64 rows per wavefront, and whatever does not fit is cut:
        // first the stage down to 512 rows per wavefront, then the window.
        const int64_t total = (int64_t)cvr::kLdsBytes / vs;
        const int64_t fixed = (int64_t)pp.wpb * cvr::kLanes + cvr::kDictMax + 4 + ((pp.hub_n + 3) & ~(int64_t)3);     // dictionary room is reserved before it is known
        int64_t stage = std::min<int64_t>(std::max<int64_t>((pp.max_nseg + 63) / 64 * 64, 64), cvr::kYStageMax);
        if (fixed + pp.wpb * stage + pp.win > total) stage = std::max<int64_t>(std::min<int64_t>(stage, 512), ((total - fixed - pp.win) / pp.wpb) & ~(int64_t)63);
        if (stage < 64) stage = 64;
        if (fixed + pp.wpb * stage + pp.win > total) pp.win = std::max<int64_t>(0, total - fixed - pp.wpb * stage) & ~(int64_t)3;
        pp.stage = stage;
    }
    return hipSuccess;
}

// A second stream per device, shared by all handles of the process, for the few analysis / conversion kernels that do not
// depend on each other (layout probe | dictionary scan, conversion | window choice): each of them is too small to fill the
// GPU and bound by latency, so side by side they take the time of one.  Created on first use (creating a stream costs
// milliseconds), never destroyed.
static hipStream_t side_stream(int device)
{
    static std::mutex  mu;
    static hipStream_t streams[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); streams[device] = nullptr; }
    return streams[device];
}

// The handle's own stream comes from a small per-device pool: creating a stream takes 2-3 ms and destroying one about as long,
// more than the whole analysis and conversion of a web-Google-sized matrix.  A stream goes back idle (cvr_destroy synchronises
// it first); at most eight are kept per device.
static std::mutex               g_pool_mu;
static std::vector<hipStream_t> g_stream_pool[64];

static hipError_t acquire_stream(int device, hipStream_t *out)
{
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_stream_pool[device].empty()) { *out = g_stream_pool[device].back(); g_stream_pool[device].pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

static void release_stream(int device, hipStream_t s)
{
    if (!s) return;
    if (device >= 0 && device < 64 && hipStreamSynchronize(s) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_stream_pool[device].size() < 8) { g_stream_pool[device].push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}

constexpr size_t kSmallProbe = 0, kSmallDictTab = 16 << 10, kSmallDictFlags = 24 << 10, kSmallBytes = 32 << 10;
constexpr size_t kPinnedProbe = 0, kPinnedDictTab = 16 << 10, kPinnedDictFlags = 24 << 10, kPinnedSmall = 32 << 10;      // in front of the planner's part of the pinned buffer

// the dictionary scan of the values [nz0, nz1) of a part, enqueued on the handle's stream: table and flags come back into
// tab_host / flags_host once the stream is synchronised
static hipError_t enqueue_dict_scan(cvr_handle *h, const void *d_va, int64_t nz0, int64_t nz1, bool f32, bool first, unsigned long long *tab_host, uint32_t *flags_host, bool last,
                                    hipStream_t st)
{
    unsigned long long *d_tab = reinterpret_cast<unsigned long long *>(h->d_small + kSmallDictTab);
    uint32_t           *d_flags = reinterpret_cast<uint32_t *>(h->d_small + kSmallDictFlags);
    hipError_t          e = hipSuccess;
    if (first && !h->small_clean) {        // (cvr_create left the table and the flags ready for the first scan)
        e = hipMemsetAsync(d_tab, 0xff, sizeof(unsigned long long) * 1024, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, sizeof(uint32_t) * 2, st);
    }
    if (e == hipSuccess) e = cvr::launch_dict_scan(d_va, nz0, nz1, f32, d_tab, d_flags, st);
    if (e == hipSuccess && last) {
        e = hipMemcpyAsync(tab_host, d_tab, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(flags_host, d_flags, sizeof(uint32_t) * 2, hipMemcpyDeviceToHost, st);
    }
    return e;
}

// The automatic layout (every layout option left at its default; one image, no column panels).  Matrices small enough
// for all their chunks to be resident at once -- 6 to 8 chunks per workgroup, one workgroup per CU -- run in the "resident"
// layout when it pays: the workgroup's chunks share an LDS window of x if a sizeable share of the non-zeros lies near the
// diagonal, and feed their rows column phase by column phase when x is larger than what an L2 keeps beside the matrix
// stream (profiles/r02_wg_window_sweep.log, r02_column_phases_sweep.log: 32.5 -> 23.4 us on the web-Google shape).
// Decided from a device-side pass over the uploaded CSR (sortedness of the rows, near-diagonal share).
static int auto_layout(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, IOpt &opt)
{
    const int64_t nnz = nz1 - nz0;
    opt.layout_auto_resident = 0;
    if (opt.steps_per_chunk != 0 || opt.waves_per_block != 0 || opt.x_window >= 0 || opt.col_phases >= 0 || opt.debug_col_mask || getenv("CVR_NO_AUTO_LAYOUT")) {
        if (opt.col_phases < 0) opt.col_phases = 0;
        return CVR_OK;
    }
    opt.col_phases = 0;
    if (nrows < 4096 || ncols < 4096) return CVR_OK;
    const int64_t vs = f32 ? 4 : 8;
    const double  slots = ((double)nnz + (double)nrows / 4) * 1.006;      // pad slots of empty rows, chunk tails
    // candidates: 8, 7 or 6 chunks per workgroup with the smallest S that keeps the workgroups at or under 252; the one
    // that fills the 256 CUs best wins (ties: more waves)
    int best_w = 0, best_S = 0;
    double best_fill = 0;
    for (int w = 8; w >= 6; w--) {
        int S = (int)std::ceil(slots / (64.0 * w * 252.0) / 4.0) * 4;
        if (S < 24 || S > 128) continue;                       // tiny shards and matrices beyond one resident pass keep the plain layout
        const double wgs = std::ceil(slots / (64.0 * w * S));
        if (wgs > best_fill) { best_fill = wgs; best_w = w; best_S = S; }
    }
    if (!best_w) return CVR_OK;
    const int64_t win = (64 * 1024) / vs;                       // 64 KiB of x per workgroup
    static_assert(sizeof(unsigned long long) * 2 * cvr::kProbeBlocks <= kSmallDictTab, "probe output fits its part of the small scratch");
    unsigned long long *d_out = reinterpret_cast<unsigned long long *>(h->d_small + kSmallProbe);
    std::vector<unsigned long long> pageable;
    const bool          pin = h->plan_ws.pinned && h->plan_ws.pinned_bytes >= kPinnedSmall;
    if (!pin) pageable.resize(2 * cvr::kProbeBlocks + 1024 + 1);
    unsigned long long *outv = pin ? reinterpret_cast<unsigned long long *>(h->plan_ws.pinned + kPinnedProbe) : pageable.data();
    unsigned long long *tabv = pin ? reinterpret_cast<unsigned long long *>(h->plan_ws.pinned + kPinnedDictTab) : pageable.data() + 2 * cvr::kProbeBlocks;
    uint32_t           *flagv = pin ? reinterpret_cast<uint32_t *>(h->plan_ws.pinned + kPinnedDictFlags) : reinterpret_cast<uint32_t *>(pageable.data() + 2 * cvr::kProbeBlocks + 1024);
    HIP_TRY(hipStreamSynchronize(h->stream));          // the upload
    const double tp0 = now_s();
    hipError_t e = cvr::launch_probe(part.d_rp, part.d_ci, nrows, ncols, (uint32_t)(win / 4), d_out, h->stream, h->small_clean);
    if (e == hipSuccess) e = hipMemcpyAsync(outv, d_out, sizeof(unsigned long long) * 2 * cvr::kProbeBlocks, hipMemcpyDeviceToHost, h->stream);
    // the dictionary scan of the values rides along: it depends on nothing decided here, and a second submission with its own
    // synchronisation costs more than the scan
    const bool with_dict = opt.value_dict != 0 && nnz > 0;
    hipStream_t side = with_dict ? side_stream(h->device) : nullptr;       // (the upload is complete: nothing to order between the two streams)
    if (!side) side = h->stream;
    if (e == hipSuccess && with_dict) e = enqueue_dict_scan(h, part.d_va, nz0, nz1, f32, true, tabv, flagv, true, side);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && side != h->stream) e = hipStreamSynchronize(side);
    unsigned long long out[2] = {0, 0};
    for (uint32_t b = 0; b < cvr::kProbeBlocks; b++) { out[0] |= outv[2 * b]; out[1] += outv[2 * b + 1]; }
    h->small_clean = false;
    if (e == hipSuccess && with_dict) { h->dict_tab.assign(tabv, tabv + 1024); h->dict_flags[0] = flagv[0]; h->dict_flags[1] = flagv[1]; h->dict_scanned = true; }
    h->info.probe_s = now_s() - tp0;
    if (e != hipSuccess) return fail(CVR_ERR_HIP, "layout probe: %s", hipGetErrorString(e));
    const bool   sorted = out[0] == 0;
    const double near = (double)out[1] / std::max<double>((double)nnz, 1.0);
    const double xbytes = (double)ncols * vs;
    // (a matrix with nearly everything near the diagonal is a band: consecutive rows share their lines of x in L1 already)
    const bool   want_win = near >= 0.15 && near < 0.9, want_phases = sorted && xbytes > 2.5e6 && near < 0.9;
    h->info.near_diagonal_share = near;
    if (!want_win && !want_phases) return CVR_OK;
    opt.layout_auto_resident = 1;
    opt.waves_per_block = best_w;
    opt.steps_per_chunk = best_S;
    opt.x_window = want_win ? (int32_t)win : 0;
    opt.col_phases = want_phases ? (int32_t)std::min(32.0, std::max(2.0, std::floor(xbytes / 600e3 + 0.5))) : 1;
    return CVR_OK;
}

// Hub table (cvr_hub.hip): hub_table > 0 asks for that many entries, < 0 decides: matrices too large for the resident layout
// whose x does not fit an L2 get the columns counted on the device, and the table is used when the columns that fit the LDS
// (beside 8 chunks' row stages) hold at least half of the (sampled) non-zeros -- R-MAT scale 22 fp32: 0.56, 402 -> 289 us;
// fp64 (half as many entries fit): 0.41, where the table loses (profiles/r02_hub_table_rmat.log).  The workgroup then has 8 chunks.
static int choose_hubs(cvr_handle *h, Part &part, const int32_t *d_ci, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, IOpt &opt, PartPlan &pp,
                       bool allow_reorder)
{
    if (opt.hub_table == 0 || opt.layout_auto_resident || opt.col_phases > 1 || nrows <= 0 || ncols >= (int64_t)cvr::kHubBit) return CVR_OK;
    const int64_t vs = f32 ? 4 : 8, nnz = nz1 - nz0;
    const bool    automatic = opt.hub_table < 0;
    if (automatic && (opt.waves_per_block != 0 || opt.x_window > 0 || opt.debug_col_mask || getenv("CVR_NO_AUTO_LAYOUT") || (double)ncols * vs < 6e6 || nnz < (8 << 20))) return CVR_OK;
    const int     wpb = opt.waves_per_block > 0 ? std::min(opt.waves_per_block, cvr::kMaxWavesPerBlock) : 8;
    const int64_t win = std::max(opt.x_window, 0);
    int64_t       room = ((int64_t)cvr::kLdsBytes / vs - (int64_t)wpb * (cvr::kLanes + 512) - cvr::kDictMax - win - 8) & ~(int64_t)1023;      // 512 staged row sums per chunk
    if (room < 1024) return automatic ? CVR_OK : fail(CVR_ERR_INVALID, "hub_table: no LDS left beside %d chunks per workgroup and the x window", wpb);
    if (!automatic) room = std::min<int64_t>(room, opt.hub_table);
    HIP_TRY(hipStreamSynchronize(h->stream));          // the upload
    const double t0 = now_s();
    cvr::HubSelection sel;
    // The whole of x re-ordered by popularity (every column index of the image becomes its rank, x_perm = x[perm] is built
    // before every SpMV): the popular columns then share cache lines and stay in the L2s.  R-MAT-22 fp64 (x = 33.5 MB): plain
    // 487 us, table alone 538, table + re-ordered x 400 us; fp32 (x = 16.8 MB): 288 -> 291 us, so only for a large x; not
    // inside column panels (a panel ranks its own range).  hub_reorder: < 0 = this rule, 0 off, 1 on.
    const bool        full_order = allow_reorder && (opt.hub_reorder > 0 || (opt.hub_reorder < 0 && (double)ncols * vs >= 24e6));
    const hipError_t  e = cvr::select_hubs(d_ci, nz0, nz1, ncols, (uint32_t)room, &sel, h->stream, full_order);
    h->info.hub_select_s += now_s() - t0;
    if (e != hipSuccess) { cvr::free_hubs(sel); return fail(CVR_ERR_HIP, "hub selection: %s", hipGetErrorString(e)); }
    h->info.hub_share = std::max(h->info.hub_share, sel.share);
    if (sel.H == 0 || (automatic && sel.share < (full_order ? 0.3 : 0.5))) { cvr::free_hubs(sel); return CVR_OK; }
    part.img.hub_n = sel.H; part.img.hub_cols = sel.hub_cols; part.img.hub_index = sel.hub_index; part.img.hub_bitmap = sel.hub_bitmap;
    part.img.order_n = sel.order_n;
    HIP_TRY(hipMalloc(&part.img.hub_x, (size_t)vs * (sel.order_n ? ((size_t)sel.order_n + 8) : ((sel.H + 3u) & ~3u))));
    pp.hub_n = sel.H;
    if (opt.waves_per_block == 0) opt.waves_per_block = wpb;
    return CVR_OK;
}

// device side of one image: allocations and uploads for a planned part (pp = nullptr: plan here, timed into *plan_s)
// (rp == nullptr: part.d_rp is already in place -- the row pointers of a column panel split on the device -- and `dr` describes it)
static int build_part(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *ci, const void *va,
                      hipMemcpyKind civa_kind, bool f32, const IOpt &opt, double *plan_s, PartPlan *planned = nullptr, const DevRows *dr = nullptr)
{
    const int64_t nz0 = rp ? (nrows ? rp[0] : 0) : dr->nz0, nz1 = rp ? (nrows ? rp[nrows] : 0) : dr->nz1;
    const size_t  vsz = f32 ? 4 : 8, nnz_span = (size_t)nz1;   // arrays are indexed literally from 0
    // the CSR goes to the device first (asynchronously): the automatic layout choice below looks at it there
    const bool adopted = rp && part.d_rp && part.d_ci && part.d_va;      // cvr_create's staging copy of the whole CSR, handed over
    if (rp && !adopted) HIP_TRY(hipMalloc(&part.d_rp, sizeof(int64_t) * ((size_t)nrows + 1)));
    if (!adopted) HIP_TRY(hipMalloc(&part.d_ci, sizeof(int32_t) * std::max<size_t>(nnz_span, 1)));
    if (!adopted) HIP_TRY(hipMalloc(&part.d_va, vsz * std::max<size_t>(nnz_span, 1)));
    if (rp && !adopted && nrows > 0) HIP_TRY(hipMemcpyAsync(part.d_rp, rp, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice, h->stream));
    if (nnz_span && !adopted) {
        HIP_TRY(hipMemcpyAsync(part.d_ci, ci, sizeof(int32_t) * nnz_span, civa_kind, h->stream));
        HIP_TRY(hipMemcpyAsync(part.d_va, va, vsz * nnz_span, civa_kind, h->stream));
    }
    PartPlan    local;
    IOpt        popt = opt;
    if (!planned) {
        int rc = auto_layout(h, part, nrows, ncols, f32, nz0, nz1, popt);      // (waits for the upload; its own pass is timed into info.probe_s)
        if (rc) return rc;
        rc = choose_hubs(h, part, part.d_ci, nrows, ncols, f32, nz0, nz1, popt, local, true);
        if (rc) return rc;
        const double   t0 = now_s();
        DevRows        here{part.d_rp, nz0, nz1, h->stream, &h->plan_ws};
        const bool     on_dev = rp && nrows > 0 && nrows >= device_plan_rows() && !getenv("CVR_HOST_PLAN");      // (the upload of row_ptr is in front of the planner's kernels on the stream)
        const int64_t *prp = on_dev ? nullptr : rp;
        const DevRows *pdr = on_dev ? &here : dr;
        HIP_TRY(plan_part(local, nrows, ncols, f32, prp, popt, pdr));
        // the resident layout wants every workgroup on a CU of its own at once: one more step per chunk until they fit
        while (popt.layout_auto_resident && !local.too_large && (int64_t)local.plan.chunks.size() > (int64_t)local.wpb * 256 && popt.steps_per_chunk < 4096) {
            popt.steps_per_chunk += 4;
            local = PartPlan();
            HIP_TRY(plan_part(local, nrows, ncols, f32, prp, popt, pdr));      // (the resident layout has no hub table: nothing of `local` to keep)
        }
        if (plan_s) *plan_s += now_s() - t0;
        planned = &local;
    }
    PartPlan &pp = *planned;
    if (pp.too_large) return fail(CVR_ERR_INVALID, "matrix too large for 32-bit row ordinals on one GPU");
    const int        S = pp.S;
    const cvr::Plan &plan = pp.plan;
    const int64_t    nchunks = (int64_t)plan.chunks.size(), yext = pp.yext;
    const std::vector<uint32_t> &desc = pp.desc, &pad = pp.pad;
    const std::vector<int64_t>  &nzb = pp.nzb;

    part.nrows = nrows; part.nnz = nz1 - nz0; part.nnz_span = nz1; part.nchunks = nchunks; part.nshared = (int64_t)plan.shared.size(); part.yext = yext;
    const int G = S / 4;
    cvr::DeviceImage &img = part.img;
    img.S = S; img.G = G; img.f32 = f32; img.nchunks = (uint32_t)nchunks; img.nrows = (uint32_t)nrows;
    img.pad_col = (uint32_t)ncols; img.nshared = (uint32_t)plan.shared.size();
    img.xcd_swizzle = opt.xcd_swizzle < 0 ? 1 : opt.xcd_swizzle > 2 ? 1 : opt.xcd_swizzle;
    img.stream_ahead = opt.stream_ahead >= 2 ? 3 : 1;
    img.depth = opt.gather_depth == 2 ? 2 : 1;
    img.wpb = (uint32_t)pp.wpb;
    img.ystage = (uint32_t)pp.stage;
    img.phases = (uint32_t)pp.phases;
    if (pp.phases > 1) {
        const int64_t pw = ((ncols + pp.phases - 1) / pp.phases + 15) / 16 * 16;
        img.phase_width = (uint32_t)std::max<int64_t>(pw, 16);
        img.col_bits = (uint32_t)pp.col_bits;
        img.tag16 = pp.tag16;
        // pieces: a lane that sits on a long row's segment falls behind the column ranges the other lanes have moved on to; with
        // chunks longer than a few steps per phase the segments are cut (auto: 8 elements once a phase takes 8 steps or more)
        if (S / pp.phases >= 8 && !opt.panel_on_one_xcd && !getenv("CVR_NO_PACE")) {      // long chunks: the SpMV kernel paces its wavefronts through the phases
            HIP_TRY(hipMalloc(&img.pace, sizeof(uint32_t) * cvr::pace_words((uint32_t)pp.phases)));
            HIP_TRY(hipMemsetAsync(img.pace, 0, sizeof(uint32_t) * cvr::pace_words((uint32_t)pp.phases), h->stream));
            img.pace_epoch = new uint32_t(0);
        }
        img.piece_max = opt.piece_max > 0 ? (uint32_t)opt.piece_max : opt.piece_max < 0 && S / pp.phases >= 8 && !opt.panel_on_one_xcd ? 8u : 0u;
        img.col_mask = pp.tag16 ? cvr::kColMask : (1u << pp.col_bits) - 1u;
    }
    if (popt.col_phases > 1 && pp.lds_short && !popt.layout_auto_resident) return fail(CVR_ERR_INVALID, "col_phases: no room for at least 63 row accumulators per chunk (LDS beside %d waves per workgroup and the x window, or %d-bit column indices)", pp.wpb, pp.col_bits);
    const int64_t win = pp.win;
    img.win_elems = (uint32_t)win;
    if (opt.debug_col_mask) img.col_mask &= (uint32_t)opt.debug_col_mask & cvr::kColMask;   // profiling knob (tools/sweep.py --colmask)

    HIP_TRY(hipMalloc(&part.d_nzb, sizeof(int64_t) * ((size_t)nchunks + 1)));
    HIP_TRY(hipMalloc(&part.d_pad, sizeof(uint32_t) * std::max<size_t>((size_t)nchunks, 1)));
    HIP_TRY(hipMalloc(&img.desc, 16 * std::max<size_t>((size_t)nchunks, 1)));
    HIP_TRY(hipMalloc(&img.target, 64 * std::max<size_t>((size_t)nchunks, 1)));
    HIP_TRY(hipMalloc(&img.shared, 24 * std::max<size_t>(plan.shared.size(), 1)));
    if (pp.phases > 1) {
        HIP_TRY(hipMalloc(&img.desc2, 8 * std::max<size_t>((size_t)nchunks, 1)));
        if (nchunks) HIP_TRY(hipMemcpyAsync(img.desc2, pp.desc2.data(), sizeof(uint32_t) * pp.desc2.size(), hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipMalloc(&img.win_base, sizeof(uint32_t) * ((size_t)nchunks / img.wpb + 1)));
    HIP_TRY(hipMemsetAsync(img.win_base, 0, sizeof(uint32_t) * ((size_t)nchunks / img.wpb + 1), h->stream));
    if (nchunks) {
        HIP_TRY(hipMemcpyAsync(part.d_nzb, nzb.data(), sizeof(int64_t) * nzb.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(part.d_pad, pad.data(), sizeof(uint32_t) * pad.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(img.desc, desc.data(), sizeof(uint32_t) * desc.size(), hipMemcpyHostToDevice, h->stream));
    }
    if (!plan.shared.empty())
        HIP_TRY(hipMemcpyAsync(img.shared, plan.shared.data(), sizeof(cvr::Shared) * plan.shared.size(), hipMemcpyHostToDevice, h->stream));
    // narrow chunks (plain layout only): if every chunk spans fewer than 32 767 columns -- banded matrices -- the image stores
    // 16-bit column offsets from the chunk's smallest column: 10 instead of 12 bytes per fp64 slot of a stream-bound SpMV
    if (popt.narrow_cols != 0 && nchunks > 0 && img.wpb == 1 && img.win_elems == 0 && img.phases <= 1 && img.hub_n == 0 && !opt.debug_col_mask) {
        uint32_t *d_wide = nullptr, wide = 1;
        HIP_TRY(hipMalloc(&img.cbase, sizeof(uint32_t) * (size_t)nchunks));
        HIP_TRY(hipMalloc(&d_wide, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(d_wide, 0, sizeof(uint32_t), h->stream));
        cvr::DeviceCsr csr;
        csr.col_idx = part.d_ci; csr.nz_begin = part.d_nzb;
        hipError_t e = cvr::launch_chunk_span(img, csr, img.cbase, d_wide, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&wide, d_wide, sizeof(wide), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(d_wide);
        if (e != hipSuccess) return fail(CVR_ERR_HIP, "chunk column spans: %s", hipGetErrorString(e));
        img.c16 = wide == 0;
        if (!img.c16) { (void)hipFree(img.cbase); img.cbase = nullptr; }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));   // the host staging vectors go out of scope; the caller may free its CSR
    return CVR_OK;
}

// second half of build_part, once it is known whether the values go through a dictionary: the stream image
static int finish_part(cvr_handle *h, Part &part)
{
    cvr::DeviceImage &img = part.img;
    img.dict = h->d_dict; img.ndict = h->ndict;
    if (img.dict) img.c16 = false;                 // (the dictionary layout keeps 32-bit column words)
    part.stream_bytes = (size_t)part.nchunks * img.G * cvr::group_bytes(img.f32, h->d_dict != nullptr, img.c16, img.tag16);
    // the SpMV kernel's software pipeline issues its stream loads up to 5 groups past the end of a chunk (the buffer
    // descriptor's range check returns zeros for them); the allocation is padded by that much so that the last chunk's
    // run-ahead stays inside it whatever the hardware does with an offset beyond num_records
    const size_t slack = 8 * (size_t)cvr::group_bytes(img.f32, h->d_dict != nullptr, img.c16, img.tag16);
    if (getenv("CVR_STREAM_UNCACHED"))   // experiment: matrix image in uncached (MTYPE UC) memory, so that it cannot displace x in L2
        HIP_TRY(hipExtMallocWithFlags((void **)&img.stream, part.stream_bytes + slack, hipDeviceMallocUncached));
    else
        HIP_TRY(hipMalloc(&img.stream, part.stream_bytes + slack));
    return CVR_OK;
}

// Column panels (the remedy when x outgrows the L2s).  The columns are cut into P ranges of
// equal width; panel p keeps, for every row that has a non-zero in its range, that row's entries of the
// range (rows compacted, order inside a row kept).  cmb_ptr / cmb_idx list, for every row, where its partial sums
// will stand in the concatenated y_ext buffers of the panels.
}  // extern "C"

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

// A parallel counting sort of the non-zeros by panel: row blocks are counted, then filled, by T host threads.
template <typename V>
static void split_panels_t(const cvr_csr_view &v, int P, PanelSplit &out)
{
    const int64_t nrows = v.nrows, ncols = v.ncols, nz0 = nrows ? v.row_ptr[0] : 0, nz1 = nrows ? v.row_ptr[nrows] : 0;
    const V      *vals = static_cast<const V *>(v.vals);
    // panels are column ranges of equal width: what has to fit the L2 is the panel's slice of x, and the panels run
    // one after the other on the whole GPU, so their non-zero counts need not balance
    const int64_t width = (ncols + P - 1) / P > 0 ? (ncols + P - 1) / P : 1;
    auto          panel_of_col = [width](int32_t c) { return (int)(c / width); };
    int T = (int)std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if (T < 1 || nz1 - nz0 < (1 << 20)) T = 1;
    std::vector<int64_t> lo((size_t)T + 1);
    for (int t = 0; t <= T; t++) lo[(size_t)t] = nrows * t / T;
    // counts per (thread, panel): non-zeros and sub-rows
    std::vector<int64_t> cn((size_t)T * P, 0), cr((size_t)T * P, 0);
    auto count = [&](int t) {
        std::vector<int64_t> last((size_t)P, -1);
        for (int64_t r = lo[(size_t)t]; r < lo[(size_t)t + 1]; r++)
            for (int64_t j = v.row_ptr[r]; j < v.row_ptr[r + 1]; j++) {
                const int p = panel_of_col(v.col_idx[j]);
                cn[(size_t)t * P + p]++;
                if (last[(size_t)p] != r) { last[(size_t)p] = r; cr[(size_t)t * P + p]++; }
            }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(count, t);
        count(0);
        for (auto &x : th) x.join();
    }
    out.rp.resize((size_t)P); out.ci.resize((size_t)P); out.va.resize((size_t)P); out.rows.resize((size_t)P);
    std::vector<int64_t> on((size_t)T * P), orow((size_t)T * P);
    for (int p = 0; p < P; p++) {
        int64_t an = 0, ar = 0;
        for (int t = 0; t < T; t++) { on[(size_t)t * P + p] = an; orow[(size_t)t * P + p] = ar; an += cn[(size_t)t * P + p]; ar += cr[(size_t)t * P + p]; }
        out.ci[(size_t)p].alloc((size_t)an);
        out.va[(size_t)p].alloc((size_t)((an * (int64_t)sizeof(V) + 7) / 8));
        out.rows[(size_t)p].alloc((size_t)ar);
        out.rp[(size_t)p].alloc((size_t)ar + 1);
        out.rp[(size_t)p][(size_t)ar] = an;
    }
    auto fill = [&](int t) {
        std::vector<int64_t> last((size_t)P, -1), pn((size_t)P), pr((size_t)P);
        for (int p = 0; p < P; p++) { pn[(size_t)p] = on[(size_t)t * P + p]; pr[(size_t)p] = orow[(size_t)t * P + p]; }
        for (int64_t r = lo[(size_t)t]; r < lo[(size_t)t + 1]; r++)
            for (int64_t j = v.row_ptr[r]; j < v.row_ptr[r + 1]; j++) {
                const int p = panel_of_col(v.col_idx[j]);
                if (last[(size_t)p] != r) {
                    last[(size_t)p] = r;
                    out.rows[(size_t)p][(size_t)pr[(size_t)p]] = (uint32_t)r;
                    out.rp[(size_t)p][(size_t)pr[(size_t)p]] = pn[(size_t)p];
                    pr[(size_t)p]++;
                }
                out.ci[(size_t)p][(size_t)pn[(size_t)p]] = v.col_idx[j];
                reinterpret_cast<V *>(out.va[(size_t)p].data())[pn[(size_t)p]] = vals[j];
                pn[(size_t)p]++;
            }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(fill, t);
        fill(0);
        for (auto &x : th) x.join();
    }
}

// Which share of the x gathers would miss a 4-MiB L2?  Eight evenly spaced windows of 65 536 consecutive rows (what one
// XCD works on at a time is of that order): in each, the gathers are counted per 128-byte line of x; the 32 768 most
// used lines (4 MiB) are taken as resident, every other gather and every first touch of a line as a miss.  The windows
// are weighted by their non-zeros.  Banded matrices: ~0; R-MAT's hub columns keep it low (scale 22, fp64: 0.13) until the
// tail outgrows the cache (scale 24, fp32: 0.22); scattered columns with little re-use: 0.44 (LiveJournal shape).
// profiles/r01_panel_rule_l2_estimate.log
static double l2_miss_estimate(const cvr_csr_view &v)
{
    const int64_t nrows = v.nrows, W = std::min<int64_t>(65536, nrows);
    if (W <= 0) return 0.0;
    const int64_t per_line = v.is_f32 ? 32 : 16, nlines = v.ncols / per_line + 1;
    const size_t  resident = (size_t)(4u << 20) / 128;
    const int     nwin = nrows == W ? 1 : 8;
    std::vector<double> refs_w((size_t)nwin, 0.0), miss_w((size_t)nwin, 0.0);
    // with the arrays on the device only the windows' slices of col_idx are fetched (row_ptr is a host copy by now)
    std::vector<std::vector<int32_t>> fetched((size_t)nwin);
    std::vector<const int32_t *>      base((size_t)nwin, nullptr);
    std::vector<int64_t>              shift((size_t)nwin, 0);
    for (int w = 0; w < nwin; w++) {
        const int64_t r0 = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
        const int64_t j0 = v.row_ptr[r0], j1 = v.row_ptr[r0 + W];
        if (v.arrays_on_device) {
            fetched[(size_t)w].resize((size_t)std::max<int64_t>(j1 - j0, 1));
            if (j1 > j0 && hipMemcpy(fetched[(size_t)w].data(), v.col_idx + j0, sizeof(int32_t) * (size_t)(j1 - j0), hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
            base[(size_t)w] = fetched[(size_t)w].data();
            shift[(size_t)w] = j0;
        } else {
            base[(size_t)w] = v.col_idx;
        }
    }
    auto window = [&](int w) {          // one thread per window, each with its own counters
        const int64_t r0 = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
        const int64_t j0 = v.row_ptr[r0], j1 = v.row_ptr[r0 + W];
        if (j1 <= j0) return;
        const int32_t *col = base[(size_t)w];
        const int64_t  sh = shift[(size_t)w];
        std::vector<uint32_t> cnt((size_t)nlines, 0u), touched;
        for (int64_t j = j0; j < j1; j++) {
            const size_t l = (size_t)(col[j - sh] / per_line);
            if (cnt[l]++ == 0) touched.push_back((uint32_t)l);
        }
        std::vector<uint32_t> top(touched.size());
        for (size_t i = 0; i < touched.size(); i++) top[i] = cnt[touched[i]];
        const size_t k = std::min(resident, top.size());
        if (k < top.size()) std::nth_element(top.begin(), top.begin() + (ptrdiff_t)k, top.end(), std::greater<uint32_t>());
        double hits = 0;
        for (size_t i = 0; i < k; i++) hits += (double)top[i] - 1.0;        // all but the first touch of a resident line
        refs_w[(size_t)w] = (double)(j1 - j0);
        miss_w[(size_t)w] = (double)(j1 - j0) - hits;
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nwin; w++) th.emplace_back(window, w);
    window(0);
    for (auto &t : th) t.join();
    double refs_all = 0, miss_all = 0;
    for (int w = 0; w < nwin; w++) { refs_all += refs_w[(size_t)w]; miss_all += miss_w[(size_t)w]; }
    return refs_all > 0 ? miss_all / refs_all : 0.0;
}

// the same estimate from a CSR in device memory (cvr_split.hip: l2_hits_device): same windows, same integers
static hipError_t l2_miss_estimate_dev(const int64_t *rp_dev, const int32_t *ci_dev, int64_t nrows, int64_t ncols, bool f32, hipStream_t st, double *miss)
{
    *miss = 0.0;
    const int64_t W = std::min<int64_t>(65536, nrows);
    if (W <= 0) return hipSuccess;
    const int nwin = nrows == W ? 1 : 8;
    int64_t   r0[8];
    double    refs[8], hits[8];
    for (int w = 0; w < nwin; w++) r0[w] = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
    const hipError_t e = cvr::l2_hits_device(rp_dev, ci_dev, r0, nwin, W, ncols, f32, (size_t)(4u << 20) / 128, refs, hits, st);
    if (e != hipSuccess) return e;
    double refs_all = 0, miss_all = 0;
    for (int w = 0; w < nwin; w++) { refs_all += refs[w]; miss_all += refs[w] - hits[w]; }
    *miss = refs_all > 0 ? miss_all / refs_all : 0.0;
    return hipSuccess;
}

static int panels_from_miss(double xb, double miss) { return miss > 0.17 ? std::min(64, std::max(2, (int)(xb * miss / 1.8e6 + 0.5))) : 1; }

static int auto_panels(const cvr_csr_view &v, double *miss_out)
{
    const double xb = (double)v.ncols * (v.is_f32 ? 4.0 : 8.0);
    int          P = 1;
    double       miss = 0;
    if (xb >= 24e6) {
        miss = l2_miss_estimate(v);
        P = panels_from_miss(xb, miss);
    }
    if (miss_out) *miss_out = miss;
    return P;
}

static void split_panels(const cvr_csr_view &v, int P, PanelSplit &out)
{
    if (v.is_f32) split_panels_t<float>(v, P, out); else split_panels_t<double>(v, P, out);
}

extern "C" {

namespace {
// CVR_CREATE_TIMING=1: wall time of the phases of cvr_create on stderr (diagnostics only)
struct PhaseClock {
    bool   on = getenv("CVR_CREATE_TIMING") && atoi(getenv("CVR_CREATE_TIMING"));
    double t = now_s();
    void   lap(const char *what) { if (on) { const double n = now_s(); fprintf(stderr, "[cvr_create] %-28s %8.2f ms\n", what, (n - t) * 1e3); t = n; } }
};
}  // namespace

int cvr_create(cvr_handle **out, const cvr_csr_view *csr_in, const cvr_options *opt_in)
{
    PhaseClock clk;
    if (!out) return fail(CVR_ERR_INVALID, "out is null");
    *out = nullptr;
    Range range("cvr_create (validate, plan, upload)");
    if (!csr_in) return fail(CVR_ERR_INVALID, "null or negative-size CSR view");
    IOpt opt = make_iopt(opt_in);
    const int ndev = cvr_device_count();
    const bool on_device = csr_in->arrays_on_device != 0;
    int rc = on_device ? CVR_OK : check_csr(csr_in);          // host arrays: rejected before any device work
    if (rc) return rc;
    clk.lap("options, check_csr (host)");
    if (ndev <= 0) return fail(CVR_ERR_NO_DEVICE, "no HIP device visible: libcvr_amd has no CPU fallback");
    if (opt.device < 0 || opt.device >= ndev) return fail(CVR_ERR_NO_DEVICE, "device %d out of range [0, %d)", opt.device, ndev);
    if (opt.steps_per_chunk != 0 && (opt.steps_per_chunk < 4 || opt.steps_per_chunk % 4 || opt.steps_per_chunk > 4096))
        return fail(CVR_ERR_INVALID, "steps_per_chunk must be a multiple of 4 in [4, 4096]");

    // CSR arrays already in device memory (of opt.device): row_ptr comes back once for the argument checks (8 B per row);
    // col_idx and vals stay where they are and are copied device to device; the chunk plan (from 200 000 rows on), the panel
    // rule and the panel split run on the device arrays (cvr_plan_dev.hip, cvr_split.hip).
    cvr_csr_view          hostv = *csr_in;
    const cvr_csr_view   *csr = &hostv;
    std::vector<int64_t>  rp_host;
    hipMemcpyKind         civa_kind = hipMemcpyHostToDevice;
    if (on_device) {
        if (hostv.nrows < 0 || hostv.ncols < 0 || (hostv.nrows > 0 && !hostv.row_ptr)) return fail(CVR_ERR_INVALID, "null or negative-size CSR view");
        HIP_TRY(hipSetDevice(opt.device));
        rp_host.resize((size_t)hostv.nrows + 1, 0);
        if (hostv.nrows > 0) HIP_TRY(hipMemcpy(rp_host.data(), hostv.row_ptr, sizeof(int64_t) * rp_host.size(), hipMemcpyDeviceToHost));
        hostv.row_ptr = rp_host.data();
        rc = check_csr(&hostv, false);
        if (rc) return rc;
        const int64_t j0 = rp_host.front(), j1 = rp_host.back();
        rc = check_columns_device(hostv.col_idx, j0, j1, hostv.ncols);
        if (rc) return rc;
        const double xb = (double)hostv.ncols * (hostv.is_f32 ? 4.0 : 8.0);
        civa_kind = hipMemcpyDeviceToDevice;
        if (opt.col_panels < 0 && !(xb >= 24e6 && j1 > 0)) opt.col_panels = 1;      // (else: the panel rule runs on the device arrays below)
        // (column panels of device arrays are split on the device: cvr_split.hip)
    }

    const int64_t nrows = csr->nrows, ncols = csr->ncols;
    const bool    f32 = csr->is_f32 != 0;
    const size_t  vsz = f32 ? 4 : 8;
    // column panels: asked for, or (col_panels < 0: auto) when x is several times the 4-MiB L2 of an XCD AND a sizeable
    // share of the gathers would miss such an L2 (l2_miss_estimate: a banded matrix re-uses its few lines, R-MAT's hub
    // columns stay resident).  One panel per 1.8 MB of *missing* x: LiveJournal shape (38.8 MB, 0.44) -> 9, the best of
    // 4..32 (profiles/r01_column_panels_livejournal_sweep2.log); R-MAT-24 fp32 (67 MB, 0.22) -> 8, 1 660 against
    // 2 300 us as one image (profiles/r01_column_panels_rmat24_fp32.log); R-MAT-22 fp64 (33.5 MB, 0.13) and matrices
    // whose x nearly fits (web-Google: profiles/r01_column_panel_probe.log) stay whole.
    const bool panels_auto = !(opt_in && opt_in->col_panels >= 0);
    clk.lap("device arrays: row_ptr, checks");
    if (nrows >= (int64_t)0xfffffff0u) return fail(CVR_ERR_INVALID, "matrix too large for 32-bit row ordinals on one GPU");

    cvr_handle *h = new (std::nothrow) cvr_handle;
    if (!h) return fail(CVR_ERR_NOMEM, "out of host memory");
    h->device = opt.device;
    h->vsz = vsz;
    cvr_info &in = h->info;
    in.nrows = nrows; in.ncols = ncols; in.nnz = nrows ? csr->row_ptr[nrows] - csr->row_ptr[0] : 0; in.is_f32 = f32 ? 1 : 0;
    in.x_elems = ncols + 1;
#define CREATE_TRY(expr)                                                                                    \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);          \
            cvr_destroy(h);                                                                                 \
            return CVR_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)
    CREATE_TRY(hipSetDevice(h->device));
    CREATE_TRY(acquire_stream(h->device, &h->stream));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->plan_ws.pinned), h->plan_ws.pinned_bytes = nrows >= device_plan_rows() ? (size_t)704 << 10 : kPinnedSmall, hipHostMallocDefault));
    CREATE_TRY(hipMalloc(&h->d_small, kSmallBytes));
    // the small scratch starts out as its first users want it (probe output and flags zero, dictionary table all ones), and the
    // two events of cvr_preprocess / cvr_spmv_bench exist: none of that in the timed analysis
    CREATE_TRY(hipMemsetAsync(h->d_small, 0, kSmallBytes, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_small + kSmallDictTab, 0xff, sizeof(unsigned long long) * 1024, h->stream));
    h->small_clean = true;
    if (nrows >= device_plan_rows()) {     // the device planner's scratch, sized for chunks of 16 steps or more (it grows if the plan needs more)
        const int64_t nnz0 = nrows ? csr->row_ptr[nrows] : 0, nb = nrows / cvr::kPlanRowBlock + 1;
        const size_t  want = (size_t)nrows * 6 + (size_t)nb * 16 + (size_t)nrows / 256 + (size_t)(2 * ((nnz0 + nrows) / 1024) + 3 * nb) * 96 + 8192;
        if (hipMalloc(&h->plan_ws.dev, want) == hipSuccess) h->plan_ws.dev_bytes = want; else (void)hipGetLastError();
    }
    h->events.resize(2);
    CREATE_TRY(hipEventCreate(&h->events[0]));
    CREATE_TRY(hipEventCreate(&h->events[1]));
    clk.lap("handle, stream");
    const double t_up0 = now_s();
    // Host arrays of a matrix that may get column panels (x of 24 MB or more, or panels asked for) are uploaded once, as they
    // are: the panel rule and the split run on that copy (building split arrays on the host means allocating, touching and
    // freeing another copy of the matrix there, which costs more than the PCIe transfer: LiveJournal shape 60 ms to split +
    // 130 ms to free against 20 ms to upload), and a matrix that stays whole adopts it as its device CSR.  The host split
    // stays as the fallback for matrices beyond the device split's 32-bit positions or when the copy does not fit.
    struct Staged {
        void *rp = nullptr, *ci = nullptr, *va = nullptr;
        void  release() { (void)hipFree(rp); (void)hipFree(ci); (void)hipFree(va); rp = ci = va = nullptr; }
        ~Staged() { release(); }
    } staged;
    const int64_t  sj0 = nrows ? csr->row_ptr[0] : 0, sj1 = nrows ? csr->row_ptr[nrows] : 0;
    const int64_t *rp_d = csr_in->row_ptr;
    const int32_t *ci_d = csr_in->col_idx;
    const void    *va_d = csr_in->vals;
    bool           dev_split = on_device;
    int            P = opt.col_panels;
    const double   xbytes = (double)ncols * (double)vsz;
    if (!on_device && (P > 1 || (P < 0 && xbytes >= 24e6)) && sj1 > 0 && sj1 < (int64_t)0xffffffffll && !getenv("CVR_HOST_SPLIT")) {
        if (hipMalloc(&staged.rp, sizeof(int64_t) * ((size_t)nrows + 1)) == hipSuccess && hipMalloc(&staged.ci, sizeof(int32_t) * (size_t)sj1) == hipSuccess &&
            hipMalloc(&staged.va, vsz * (size_t)sj1) == hipSuccess &&
            hipMemcpy(staged.rp, csr->row_ptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(staged.ci, csr->col_idx, sizeof(int32_t) * (size_t)sj1, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(staged.va, csr->vals, vsz * (size_t)sj1, hipMemcpyHostToDevice) == hipSuccess) {
            rp_d = static_cast<const int64_t *>(staged.rp); ci_d = static_cast<const int32_t *>(staged.ci); va_d = staged.va;
            dev_split = true;
        } else {
            (void)hipGetLastError();      // not enough device memory for the staging copy: rule and split on the host
            staged.release();
        }
    }
    clk.lap("staging upload");
    // the panel rule (col_panels < 0): on the device copy when there is one (same windows, same integers as the host form)
    const double t_rule0 = now_s();
    if (P < 0) {
        if (xbytes < 24e6 || sj1 <= sj0) P = 1;
        else if (dev_split) {
            double miss = 0;
            CREATE_TRY(l2_miss_estimate_dev(rp_d, ci_d, nrows, ncols, f32, h->stream, &miss));
            P = panels_from_miss(xbytes, miss);
        } else P = auto_panels(*csr, nullptr);
    }
    const double panel_rule_s = now_s() - t_rule0;          // part of the analysis: added to plan_s below
    clk.lap("panel rule");
    // Panels in rounds of eight, each on one XCD (run_spmv, d_multi): the count the rule gave is for slices that share every L2 in
    // turn; with an L2 to itself a slice may be ~2.6 MB, and the count is a multiple of eight (CVR_XCD_PANELS=0: the old form)
    const char *xp_env = getenv("CVR_XCD_PANELS");
    const bool  xcd_panels = !(xp_env && atoi(xp_env) == 0);
    if (P > 1 && panels_auto && xcd_panels && xp_env && atoi(xp_env) > 1) P = atoi(xp_env);          // (experiments: the count itself)
    else if (P > 1 && panels_auto && xcd_panels) P = 8 * std::max(1, (int)std::ceil(xbytes / (8.0 * 2.6e6)));
    if (P < 1) P = 1;
    if (P > 64) P = 64;
    in.col_panels = P;
    h->parts.resize((size_t)P);
    if (P == 1) {
        if (staged.rp) {          // the staging copy becomes the part's device CSR
            Part &part = h->parts[0];
            part.d_rp = static_cast<int64_t *>(staged.rp); part.d_ci = static_cast<int32_t *>(staged.ci); part.d_va = staged.va;
            staged.rp = staged.ci = staged.va = nullptr;
        }
        rc = build_part(h, h->parts[0], nrows, ncols, csr->row_ptr, csr->col_idx, csr->vals, civa_kind, f32, opt, &in.plan_s);
        if (rc) { cvr_destroy(h); return rc; }
        in.yext_elems = h->parts[0].yext;
    } else {
        PanelSplit   sp;
        struct SplitGuard { cvr::DeviceSplit d; ~SplitGuard() { cvr::free_device_split(d); } } dsg;
        const double t0 = now_s();
        // Power-law matrices whose popular columns will sit in hub tables need fewer, wider panels: the table takes the hot
        // half of the gathers off the L2s, and what remains runs best with ~16 MB of x per panel instead of ~4 (R-MAT-26 fp32
        // on one GPU: 59 panels 6.4 ms, 16 panels 5.4 ms; R-MAT-24: 8 and 4 panels alike; profiles/r02_hub_table_rmat.log)
        if (dev_split && panels_auto && opt.hub_table < 0 && opt.waves_per_block == 0 && opt.x_window <= 0 && !getenv("CVR_NO_AUTO_LAYOUT")) {      // (the predicate of choose_hubs: only panels that will get tables are widened)
            const int64_t room = ((int64_t)cvr::kLdsBytes / (int64_t)vsz - 8 * (cvr::kLanes + 512) - cvr::kDictMax - 8) & ~(int64_t)1023;
            cvr::HubSelection sel;
            const double      th0 = now_s();
            const hipError_t  e = cvr::select_hubs(ci_d, sj0, sj1, ncols, (uint32_t)std::max<int64_t>(room, 1024), &sel, h->stream);
            in.hub_select_s += now_s() - th0;
            const double share = sel.share;
            cvr::free_hubs(sel);
            if (e != hipSuccess) { cvr_destroy(h); return fail(CVR_ERR_HIP, "hub selection: %s", hipGetErrorString(e)); }
            if (share >= 0.25) {       // (of the whole matrix: the panels' own tables, ranked inside their ranges, hold more)
                const int Pw = std::max(2, (int)std::ceil((double)ncols * (double)vsz / 16e6));
                if (Pw < P) { P = Pw; h->parts.resize((size_t)P); in.col_panels = P; }
            }
            // popularity too flat for any panel's table to reach the half it needs (a panel's own top columns hold a few times
            // the whole matrix's share at most: LiveJournal shape 0.06): the panels skip their own counting passes
            if (share < 0.08) opt.hub_table = 0;
        }
        clk.lap("  panel count with hub tables");
        std::vector<int64_t> nsubs((size_t)P, 0);
        if (dev_split) {        // nothing but a few counts comes to the host: the sub-rows are planned where they are (cvr_plan_dev.hip)
            const int64_t  width = (ncols + P - 1) / P > 0 ? (ncols + P - 1) / P : 1;
            const hipError_t e = cvr::split_panels_device(rp_d, ci_d, va_d, f32, nrows, sj0, sj1, width, P, &dsg.d, h->stream);
            if (e != hipSuccess) { cvr_destroy(h); return fail(CVR_ERR_HIP, "column-panel split on the device: %s", hipGetErrorString(e)); }
            staged.release();       // the split arrays replace the staging copy
            for (int p = 0; p < P; p++) nsubs[(size_t)p] = dsg.d.sub0[p + 1] - dsg.d.sub0[p];
        } else {
            split_panels(*csr, P, sp);
            for (int p = 0; p < P; p++) nsubs[(size_t)p] = (int64_t)sp.rows[(size_t)p].size();
        }
        in.plan_s += now_s() - t0 - in.hub_select_s;      // (the hub count that sizes the panels is reported on its own)
        clk.lap("panel split");
        std::vector<PartPlan> pps((size_t)P);
        IOpt                  panel_opt = opt;
        panel_opt.col_phases = getenv("CVR_PANEL_PHASES") ? atoi(getenv("CVR_PANEL_PHASES")) : 1;          // column phases are for the single image whose chunks are all resident at once
        panel_opt.panel_on_one_xcd = xcd_panels ? 1 : 0;
        std::vector<IOpt>        popts((size_t)P, panel_opt);
        std::vector<DevRows>     drs((size_t)P);
        if (dev_split) {
            // per panel: its row pointers made panel-local (a slice of the split's, minus the panel's first position), a hub
            // table of the most popular columns of its own range, and the chunk plan -- all from device arrays
            const double tp = now_s(), hub0 = in.hub_select_s;
            for (int p = 0; p < P; p++) {
                Part         &part = h->parts[(size_t)p];
                const int64_t ns = nsubs[(size_t)p], nzp = dsg.d.off[p + 1] - dsg.d.off[p];
                CREATE_TRY(hipMalloc(&part.d_rp, sizeof(int64_t) * ((size_t)ns + 1)));
                CREATE_TRY(cvr::launch_shift_rows(dsg.d.rp + dsg.d.sub0[p], ns + 1, dsg.d.off[p], part.d_rp, h->stream));
                drs[(size_t)p] = DevRows{part.d_rp, 0, nzp, h->stream, &h->plan_ws};
                rc = choose_hubs(h, part, dsg.d.ci + dsg.d.off[p], ns, ncols, f32, 0, nzp, popts[(size_t)p], pps[(size_t)p], false);
                if (rc) { cvr_destroy(h); return rc; }
                CREATE_TRY(plan_part(pps[(size_t)p], ns, ncols, f32, nullptr, popts[(size_t)p], &drs[(size_t)p]));
            }
            in.plan_s += now_s() - tp - (in.hub_select_s - hub0);      // (hub selection is reported on its own)
            clk.lap("  hub tables, plans (device)");
        } else {
            // the panels' images are planned side by side (the host planner is a sequential walk per image), then built one by one
            const double tp = now_s();
            int T = (int)std::thread::hardware_concurrency();
            T = std::max(1, std::min(T, P));
            auto work = [&](int t) { for (int p = t; p < P; p += T) { pps[(size_t)p].plan_threads = 1; (void)plan_part(pps[(size_t)p], nsubs[(size_t)p], ncols, f32, sp.rp[(size_t)p].data(), popts[(size_t)p]); } };
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            in.plan_s += now_s() - tp;
            clk.lap("  panels planned (parallel)");
        }
        int64_t zoff = 0, nsub = 0;
        for (int p = 0; p < P; p++) {
            Part &part = h->parts[(size_t)p];
            if (dev_split)
                rc = build_part(h, part, nsubs[(size_t)p], ncols, nullptr, dsg.d.ci + dsg.d.off[p],
                                static_cast<const uint8_t *>(dsg.d.va) + (size_t)dsg.d.off[p] * vsz, hipMemcpyDeviceToDevice, f32, popts[(size_t)p], &in.plan_s, &pps[(size_t)p], &drs[(size_t)p]);
            else
                rc = build_part(h, part, nsubs[(size_t)p], ncols, sp.rp[(size_t)p].data(), sp.ci[(size_t)p].data(),
                                sp.va[(size_t)p].data(), hipMemcpyHostToDevice, f32, popts[(size_t)p], &in.plan_s, &pps[(size_t)p]);
            if (rc) { cvr_destroy(h); return rc; }
            part.zoff = zoff;
            zoff += part.yext;
            nsub += part.nrows;
        }
        if (zoff >= (int64_t)0xffffffffu) { cvr_destroy(h); return fail(CVR_ERR_INVALID, "partial-sum buffer too large for 32-bit indices"); }
        clk.lap("parts: plan, alloc, copies");
        // combine tables: the rows of every panel's sub-rows (concatenated) and, per panel, where each block of
        // kCombineRows rows starts among them
        const double   t1 = now_s();
        const uint32_t nblocks = (uint32_t)((nrows + cvr::kCombineRows - 1) / cvr::kCombineRows);
        const size_t   nboff = (size_t)P * (nblocks + 1);
        CREATE_TRY(hipEventCreateWithFlags(&h->z_free, hipEventDisableTiming));
        CREATE_TRY(hipMalloc(&h->d_z, vsz * (size_t)std::max<int64_t>(zoff, 1)));
        CREATE_TRY(hipMalloc(&h->d_block_off, sizeof(uint32_t) * nboff));
        CREATE_TRY(hipMalloc(&h->d_cpanels, sizeof(cvr::CombinePanel) * (size_t)P));
        CREATE_TRY(hipMemsetAsync(h->d_z, 0, vsz * (size_t)std::max<int64_t>(zoff, 1), h->stream));
        std::vector<cvr::CombinePanel> cps((size_t)P);
        std::vector<uint32_t>          block_off;
        if (dev_split) {
            h->d_rows = dsg.d.rows;        // the split's row numbers are the combine pass's, as they stand
            dsg.d.rows = nullptr;
            int64_t roff = 0;
            for (int p = 0; p < P; p++) {
                CREATE_TRY(cvr::launch_block_off(h->d_rows + roff, (uint32_t)nsubs[(size_t)p], nblocks, h->d_block_off + (size_t)p * (nblocks + 1), h->stream));
                cps[(size_t)p] = cvr::CombinePanel{static_cast<uint8_t *>(h->d_z) + (size_t)h->parts[(size_t)p].zoff * vsz, h->d_rows + roff};
                roff += nsubs[(size_t)p];
            }
        } else {
            block_off.resize(nboff);
            for (int p = 0; p < P; p++) {
                const Raw<uint32_t> &rows = sp.rows[(size_t)p];
                uint32_t            *bo = block_off.data() + (size_t)p * (nblocks + 1);
                size_t               u = 0;
                for (uint32_t b2 = 0; b2 <= nblocks; b2++) {
                    const uint64_t lim = (uint64_t)b2 * cvr::kCombineRows;
                    while (u < rows.size() && rows[u] < lim) u++;
                    bo[b2] = (uint32_t)u;
                }
            }
            CREATE_TRY(hipMalloc(&h->d_rows, sizeof(uint32_t) * (size_t)std::max<int64_t>(nsub, 1)));
            int64_t roff = 0;
            for (int p = 0; p < P; p++) {
                const Raw<uint32_t> &rows = sp.rows[(size_t)p];
                if (rows.size()) CREATE_TRY(hipMemcpyAsync(h->d_rows + roff, rows.data(), sizeof(uint32_t) * rows.size(), hipMemcpyHostToDevice, h->stream));
                cps[(size_t)p] = cvr::CombinePanel{static_cast<uint8_t *>(h->d_z) + (size_t)h->parts[(size_t)p].zoff * vsz, h->d_rows + roff};
                roff += (int64_t)rows.size();
            }
            CREATE_TRY(hipMemcpyAsync(h->d_block_off, block_off.data(), sizeof(uint32_t) * block_off.size(), hipMemcpyHostToDevice, h->stream));
        }
        in.plan_s += now_s() - t1;
        CREATE_TRY(hipMemcpyAsync(h->d_cpanels, cps.data(), sizeof(cvr::CombinePanel) * (size_t)P, hipMemcpyHostToDevice, h->stream));
        clk.lap("  combine tables enqueued");
        std::vector<cvr::FixPart> fp((size_t)P);
        for (int p = 0; p < P; p++) {
            const Part &part = h->parts[(size_t)p];
            fp[(size_t)p] = cvr::FixPart{part.img.shared, static_cast<uint8_t *>(h->d_z) + (size_t)part.zoff * vsz, (uint32_t)part.nshared, (uint32_t)part.nrows};
            h->max_nshared = std::max(h->max_nshared, (uint32_t)part.nshared);
        }
        CREATE_TRY(hipMalloc(&h->d_fixparts, sizeof(cvr::FixPart) * (size_t)P));
        CREATE_TRY(hipMemcpyAsync(h->d_fixparts, fp.data(), sizeof(cvr::FixPart) * (size_t)P, hipMemcpyHostToDevice, h->stream));
        CREATE_TRY(hipStreamSynchronize(h->stream));
        clk.lap("  tables synchronised");
        in.yext_elems = nrows + 1;
        in.image_bytes += (int64_t)(sizeof(uint32_t) * ((size_t)nsub + nboff));
    }
    clk.lap("combine tables / single part");
    // value dictionary (value_dict: <0 auto, 0 off): one code byte per slot instead of the value when the matrix has at
    // most 256 distinct values -- 12 -> 5 bytes per slot for fp64 (profiles/r01_value_dictionary.log).  The distinct
    // values are collected on the device from the uploaded CSR (a 40-MB scan takes microseconds there, milliseconds
    // on the host).
    const double t_dict0 = now_s();
    if (opt.value_dict != 0 && in.nnz > 0) {
        if (!h->dict_scanned) {          // (single images with the automatic layout scanned together with the layout probe)
            h->dict_tab.assign(1024, ~0ull);
            for (size_t i = 0; i < h->parts.size(); i++) {
                const Part &p = h->parts[i];
                CREATE_TRY(enqueue_dict_scan(h, p.d_va, p.nnz_span - p.nnz, p.nnz_span, f32, i == 0, h->dict_tab.data(), h->dict_flags, i + 1 == h->parts.size(), h->stream));
            }
            CREATE_TRY(hipStreamSynchronize(h->stream));
            h->small_clean = false;
        }
        const std::vector<unsigned long long> &tab = h->dict_tab;
        const uint32_t                        *flags = h->dict_flags;
        if (!(flags[0] & 1u)) {
            std::vector<unsigned long long> d;
            d.push_back(0);                                                  // +0.0: the value of every pad slot
            for (unsigned long long b : tab) if (b != ~0ull) d.push_back(b);
            if (flags[0] & 2u) d.push_back(f32 ? 0xffffffffull : ~0ull);     // the all-ones pattern occurs as a value
            std::sort(d.begin(), d.end());
            d.erase(std::unique(d.begin(), d.end()), d.end());
            if (d.size() <= (size_t)cvr::kDictMax) {
                h->ndict = (uint32_t)d.size();
                std::vector<uint32_t> d32(d.begin(), d.end());
                CREATE_TRY(hipMalloc(&h->d_dict, vsz * (size_t)cvr::kDictMax));
                CREATE_TRY(hipMemsetAsync(h->d_dict, 0, vsz * (size_t)cvr::kDictMax, h->stream));
                CREATE_TRY(hipMemcpyAsync(h->d_dict, f32 ? (const void *)d32.data() : (const void *)d.data(), vsz * h->ndict, hipMemcpyHostToDevice, h->stream));
                CREATE_TRY(hipStreamSynchronize(h->stream));
            }
        }
    }
    in.value_dict = (int32_t)h->ndict;
    in.dict_s = now_s() - t_dict0;
    clk.lap("value dictionary scan");
    for (Part &p : h->parts) { rc = finish_part(h, p); if (rc) { cvr_destroy(h); return rc; } }
    for (const Part &p : h->parts) in.hub_entries = std::max<int32_t>(in.hub_entries, (int32_t)p.img.hub_n);
    // column panels whose images are plain (one chunk per workgroup, no LDS tables): eight panels per launch, each on the XCD of
    // its workgroups, so that an L2 holds one slice of x at a time and every line of x is fetched by one XCD only
    if (h->paneled() && xcd_panels) {
        bool plain = true;
        for (const Part &p : h->parts) plain = plain && p.img.wpb <= 1 && p.img.hub_n == 0 && p.img.win_elems == 0 && p.img.phases == h->parts[0].img.phases && p.img.tag16 == h->parts[0].img.tag16 && !p.img.c16 && p.img.S == h->parts[0].img.S;
        if (plain) {
            const size_t per_round = getenv("CVR_XCD_PANELS_DEBUG") ? (size_t)atoi(getenv("CVR_XCD_PANELS_DEBUG")) : 8;      // (diagnostics: fewer panels side by side)
            const size_t rounds = (h->parts.size() + per_round - 1) / per_round;
            std::vector<cvr::PanelArgs> pa(rounds * 8, cvr::PanelArgs{nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr});
            h->multi_chunks.assign(rounds, 0u);
            for (size_t j = 0; j < h->parts.size(); j++) {
                const Part &p = h->parts[j];
                const size_t i = (j / per_round) * 8 + j % per_round;
                pa[i] = cvr::PanelArgs{p.img.stream, p.img.desc, p.img.target, static_cast<uint8_t *>(h->d_z) + (size_t)p.zoff * vsz, p.img.nchunks, p.img.ystage, p.img.desc2};
                h->multi_chunks[i / 8] = std::max(h->multi_chunks[i / 8], p.img.nchunks);
                h->multi_ystage = std::max(h->multi_ystage, p.img.ystage);
                if (getenv("CVR_XCD_PANELS_TRACE")) fprintf(stderr, "[xcd panels] part %zu slot %zu nchunks %u ystage %u S %d G %d zoff %lld yext %lld nshared %u stream %p\n", j, i, p.img.nchunks, p.img.ystage, p.img.S, p.img.G, (long long)p.zoff, (long long)p.yext, p.img.nshared, (void *)p.img.stream);
            }
            CREATE_TRY(hipMalloc(&h->d_multi, sizeof(cvr::PanelArgs) * pa.size()));
            CREATE_TRY(hipMemcpy(h->d_multi, pa.data(), sizeof(cvr::PanelArgs) * pa.size(), hipMemcpyHostToDevice));
        }
    }
    if (!h->z_free && in.hub_entries) CREATE_TRY(hipEventCreateWithFlags(&h->z_free, hipEventDisableTiming));
    in.steps_per_chunk = h->parts[0].img.S;
    in.col_phases = (int32_t)h->parts[0].img.phases; in.waves_per_block = (int32_t)h->parts[0].img.wpb; in.x_window = (int32_t)h->parts[0].img.win_elems;
    in.lds_bytes = (int32_t)cvr::spmv_lds_bytes(h->parts[0].img);
    in.narrow_cols = h->parts[0].img.c16 ? 1 : 0;
    in.hub_reorder = h->parts[0].img.order_n ? 1 : 0;
    in.row_tags16 = h->parts[0].img.tag16 ? 1 : 0;
    in.row_bands = 1;
    in.piece_max = (int32_t)h->parts[0].img.piece_max;
    in.chunk_row_cap = h->parts[0].img.phases > 1 ? (int64_t)h->parts[0].img.ystage - 1 : 0;
    for (const Part &p : h->parts) {
        in.nchunks += p.nchunks; in.nshared += p.nshared; in.nslots += p.nchunks * 64 * p.img.S;
        in.image_bytes += (int64_t)(p.stream_bytes + (size_t)p.nchunks * (16 + 64) + (size_t)p.nshared * 24);
    }
    CREATE_TRY(hipMalloc(&h->d_err, sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(&h->d_x, vsz * (size_t)in.x_elems));
    CREATE_TRY(hipMalloc(&h->d_y, vsz * (size_t)in.yext_elems));
    CREATE_TRY(hipMemsetAsync(h->d_x, 0, vsz * (size_t)in.x_elems, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_y, 0, vsz * (size_t)in.yext_elems, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
    CREATE_TRY(hipStreamSynchronize(h->stream));   // the caller may free its CSR when this returns
    in.upload_s = now_s() - t_up0 - in.plan_s - in.probe_s - panel_rule_s - in.hub_select_s;      // (hub selection and the layout probe are reported on their own)
    in.plan_s += panel_rule_s;
    cvr::free_plan_scratch(h->plan_ws);
    (void)hipFree(h->d_small); h->d_small = nullptr;
    std::vector<unsigned long long>().swap(h->dict_tab);
    clk.lap("images, x, y");
#undef CREATE_TRY
    *out = h;
    return CVR_OK;
}

int cvr_preprocess(cvr_handle *h, int keep_csr, double *seconds)
{
    if (!h) return fail(CVR_ERR_INVALID, "handle is null");
    if (h->parts.empty() || !h->parts[0].d_rp) return fail(CVR_ERR_STATE, "the device CSR was already released: cvr_preprocess runs once unless keep_csr was set");
    Range range("cvr_preprocess (CSR -> CVR64)");
    const double t_wall0 = now_s();
    HIP_TRY(hipSetDevice(h->device));
    if (h->events.size() < 2) {
        h->events.resize(2);
        HIP_TRY(hipEventCreate(&h->events[0]));
        HIP_TRY(hipEventCreate(&h->events[1]));
    }
    const hipEvent_t e0 = h->events[0], e1 = h->events[1];
    HIP_TRY(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
    struct SegGuard { cvr::SegTable t; void *arena = nullptr; ~SegGuard() { (void)hipFree(arena); } } sg;      // (one allocation: six cost six times the call)
    Part &p0 = h->parts[0];
    // column phases: the segment table (conversion-time only) gives every chunk room for as many segments as it has slots, so
    // that counting and filling are one kernel per chunk and the conversion follows without the host in between; the images of a
    // handle (column panels) are converted one after the other and share the table, sized for the largest
    size_t seg_n1 = 0, seg_chunks = 0;
    for (const Part &p : h->parts)
        if (p.img.phases > 1 && p.nchunks > 0) {
            seg_n1 = std::max(seg_n1, (size_t)p.nchunks * (size_t)cvr::kLanes * (size_t)p.img.S);
            seg_chunks = std::max(seg_chunks, (size_t)p.nchunks);
        }
    const bool phased = seg_n1 > 0;
    if (phased) {
        cvr::SegTable &t = sg.t;
        const size_t n1 = seg_n1;
        if (n1 >= ((size_t)1 << 32)) return fail(CVR_ERR_INVALID, "col_phases: more than 2^32 slots in one image");
        auto         up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_begin = 0, o_len = o_begin + up(sizeof(int64_t) * n1), o_row = o_len + up(sizeof(uint32_t) * n1), o_cnt = o_row + up(sizeof(uint16_t) * n1),
                     o_flags = o_cnt + up(sizeof(uint32_t) * (seg_chunks + 1));
        HIP_TRY(hipMalloc(&sg.arena, o_flags + 256));
        uint8_t *a = static_cast<uint8_t *>(sg.arena);
        t.begin = reinterpret_cast<int64_t *>(a + o_begin); t.len = reinterpret_cast<uint32_t *>(a + o_len); t.row = reinterpret_cast<uint16_t *>(a + o_row);
        t.cnt = reinterpret_cast<uint32_t *>(a + o_cnt); t.flags = reinterpret_cast<uint32_t *>(a + o_flags);
        HIP_TRY(hipMemsetAsync(t.flags, 0, sizeof(uint32_t) * 2, h->stream));
    }
    // the window choice needs nothing of the conversion (and nothing is pending on the handle's stream: cvr_create ended with a
    // synchronisation): a single image's runs beside it on the side stream
    hipStream_t wstream = h->paneled() || !p0.img.win_elems ? h->stream : side_stream(h->device);
    if (!wstream) wstream = h->stream;
    HIP_TRY(hipEventRecord(e0, h->stream));
    uint32_t              seg_flags[2] = {0, 0};
    std::vector<uint32_t> seg_totals(h->parts.size(), 0u);
    for (Part &p : h->parts) {
        cvr::DeviceCsr csr;
        csr.row_ptr = p.d_rp; csr.col_idx = p.d_ci; csr.vals = p.d_va; csr.nz_begin = p.d_nzb; csr.pad_cnt = p.d_pad;
        if (wstream != h->stream) HIP_TRY(cvr::launch_window(p.img, csr, wstream));
        if (p.img.phases > 1 && p.nchunks > 0) {
            cvr::SegTable &t = sg.t;
            HIP_TRY(cvr::launch_seg_build(p.img, csr, t, h->stream));
            HIP_TRY(cvr::launch_convert(p.img, csr, h->d_err, h->stream, &t));
            HIP_TRY(hipMemcpyAsync(seg_flags, t.flags, sizeof(seg_flags), hipMemcpyDeviceToHost, h->stream));      // (the flags of all images so far)
            HIP_TRY(hipMemcpyAsync(&seg_totals[(size_t)(&p - h->parts.data())], t.cnt + p.nchunks, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        } else {
            HIP_TRY(cvr::launch_convert(p.img, csr, h->d_err, h->stream));
        }
        if (wstream == h->stream) HIP_TRY(cvr::launch_window(p.img, csr, h->stream));
    }
    HIP_TRY(hipEventRecord(e1, h->stream));
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, h->d_err, sizeof(err), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (wstream != h->stream) HIP_TRY(hipStreamSynchronize(wstream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    h->info.convert_s = ms * 1e-3;
    h->info.lds_bytes = (int32_t)cvr::spmv_lds_bytes(h->parts[0].img);      // column phases: the segment-row copy is sized by now
    if (seconds) *seconds = ms * 1e-3;
    if (seg_flags[0] & 1u) return fail(CVR_ERR_INVALID, "col_phases needs the column indices of every row in ascending order");
    if (err) return fail(CVR_ERR_INTERNAL, "device converter self-check failed (flags 0x%x)", err);
    h->info.nsegments = 0;
    for (uint32_t v : seg_totals) h->info.nsegments += v;
    h->info.preprocess_wall_s = now_s() - t_wall0;
    h->converted = true;
    if (!keep_csr) for (Part &p : h->parts) p.release_csr();
    return CVR_OK;
}

int cvr_get_info(const cvr_handle *h, cvr_info *info)
{
    if (!h || !info) return fail(CVR_ERR_INVALID, "null argument");
    *info = h->info;
    return CVR_OK;
}

int cvr_destroy(cvr_handle *h)
{
    if (!h) return CVR_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (Part &p : h->parts) p.release_all();
    cvr::free_plan_scratch(h->plan_ws);
    if (h->d_small) (void)hipFree(h->d_small);
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    if (h->z_free) (void)hipEventDestroy(h->z_free);
    for (void *p : {(void *)h->d_err, h->d_z, (void *)h->d_rows, (void *)h->d_block_off, (void *)h->d_cpanels, (void *)h->d_fixparts, (void *)h->d_multi, h->d_dict, h->d_x, h->d_y}) if (p) (void)hipFree(p);
    release_stream(h->device, h->stream);
    delete h;
    return CVR_OK;
}

void *cvr_x_device(cvr_handle *h) { return h ? h->d_x : nullptr; }
void *cvr_y_device(cvr_handle *h) { return h ? h->d_y : nullptr; }
void *cvr_stream(cvr_handle *h) { return h ? (void *)h->stream : nullptr; }

int cvr_spmv_device(cvr_handle *h, const void *x_dev, void *y_dev, void *stream)
{
    if (!h || !x_dev || !y_dev) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    HIP_TRY(hipSetDevice(h->device));          // the NULL stream means the current device's
    HIP_TRY(run_spmv(h, x_dev, y_dev, (hipStream_t)stream));
    return CVR_OK;
}

int cvr_spmv_device_repeat(cvr_handle *h, const void *x_dev, void *y_dev, void *stream, int n)
{
    if (!h || !x_dev || !y_dev) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    const hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < n; i++) HIP_TRY(run_spmv(h, x_dev, y_dev, st));
    return CVR_OK;
}

// ---- the exchange step of the row-sharded SpMV: RCCL, one process per GPU -----------------------------------

}  // extern "C"

namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId)    get_unique_id = nullptr;
    decltype(&ncclCommInitRank)   comm_init_rank = nullptr;
    decltype(&ncclCommDestroy)    comm_destroy = nullptr;
    decltype(&ncclAllGather)      all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

// RCCL is half a gigabyte of code objects: load it only when a communicator is asked for, and prefer the
// instance the process already holds (PyTorch ships its own librccl.so) so that one runtime serves both.
const RcclApi *rccl_api()
{
    static const RcclApi api = [] {            // initialised once, thread-safe (C++11 function-local static)
        RcclApi a;
        const char *names[] = {getenv("CVR_RCCL_LIB"), "librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2 && !a.lib; pass++)
            for (const char *n : names) {
                if (!n || !*n) continue;
                a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (a.lib) break;
            }
        if (a.lib) {
            a.get_unique_id = (decltype(a.get_unique_id))dlsym(a.lib, "ncclGetUniqueId");
            a.comm_init_rank = (decltype(a.comm_init_rank))dlsym(a.lib, "ncclCommInitRank");
            a.comm_destroy = (decltype(a.comm_destroy))dlsym(a.lib, "ncclCommDestroy");
            a.all_gather = (decltype(a.all_gather))dlsym(a.lib, "ncclAllGather");
            a.error_string = (decltype(a.error_string))dlsym(a.lib, "ncclGetErrorString");
            if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather || !a.error_string) a.lib = nullptr;
        }
        return a;
    }();
    return api.lib ? &api : nullptr;
}

#define RCCL_TRY(api, expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, (api)->error_string(r_), __FILE__, __LINE__); \
    } while (0)

}  // namespace

struct cvr_comm {
    ncclComm_t  comm = nullptr;
    int         nranks = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;                 // the collectives of cvr_spmv_gather_repeat run here
    hipEvent_t  ready[2] = {nullptr, nullptr};    // y_dev[b] computed
    hipEvent_t  done[2] = {nullptr, nullptr};     // gather of y_dev[b] into yall_dev[b] finished
    bool        pending[2] = {false, false};
};

extern "C" {

int cvr_comm_unique_id(void *id128)
{
    if (!id128) return fail(CVR_ERR_INVALID, "null argument");
    const RcclApi *api = rccl_api();
    if (!api) return fail(CVR_ERR_NO_DEVICE, "RCCL not found (librccl.so; set CVR_RCCL_LIB): %s", dlerror());
    static_assert(sizeof(ncclUniqueId) == CVR_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RCCL_TRY(api, api->get_unique_id(&id));
    memcpy(id128, &id, sizeof(id));
    return CVR_OK;
}

int cvr_comm_destroy(cvr_comm *c);

int cvr_comm_create(cvr_comm **out, const void *id128, int nranks, int rank, int device)
{
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(CVR_ERR_INVALID, "bad communicator arguments");
    *out = nullptr;
    const RcclApi *api = rccl_api();
    if (!api) return fail(CVR_ERR_NO_DEVICE, "RCCL not found (librccl.so; set CVR_RCCL_LIB): %s", dlerror());
    HIP_TRY(hipSetDevice(device));
    cvr_comm *c = new (std::nothrow) cvr_comm;
    if (!c) return fail(CVR_ERR_NOMEM, "out of host memory");
    c->nranks = nranks; c->rank = rank; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = api->comm_init_rank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        cvr_comm_destroy(c);
        return fail(CVR_ERR_HIP, "ncclCommInitRank: %s", api->error_string(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipEventCreateWithFlags(&c->ready[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[b], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        cvr_comm_destroy(c);           // whatever was created so far
        return fail(CVR_ERR_HIP, "communicator stream / events: %s", hipGetErrorString(e));
    }
    *out = c;
    return CVR_OK;
}

int cvr_comm_destroy(cvr_comm *c)
{
    if (!c) return CVR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    const RcclApi *api = rccl_api();
    if (api && c->comm) (void)api->comm_destroy(c->comm);
    for (int b = 0; b < 2; b++) {
        if (c->ready[b]) (void)hipEventDestroy(c->ready[b]);
        if (c->done[b]) (void)hipEventDestroy(c->done[b]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CVR_OK;
}

int cvr_comm_all_gather(cvr_comm *c, const void *send_dev, void *recv_dev, int64_t count, int is_f32, void *stream)
{
    if (!c || !send_dev || !recv_dev || count < 0) return fail(CVR_ERR_INVALID, "bad all-gather arguments");
    const RcclApi *api = rccl_api();
    if (!api) return fail(CVR_ERR_NO_DEVICE, "RCCL not loaded");
    RCCL_TRY(api, api->all_gather(send_dev, recv_dev, (size_t)count, is_f32 ? ncclFloat : ncclDouble, c->comm, (hipStream_t)stream));
    return CVR_OK;
}

int cvr_spmv_gather_repeat(cvr_handle *h, cvr_comm *c, const void *x_dev, void *const y_dev[2], void *const yall_dev[2],
                           int64_t max_rows, int n, int overlap, void *stream, int *last_buf)
{
    if (!h || !c || !x_dev || !y_dev || !yall_dev || !y_dev[0] || !y_dev[1] || !yall_dev[0] || !yall_dev[1])
        return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    if (max_rows < h->info.nrows) return fail(CVR_ERR_INVALID, "max_rows %lld < the %lld rows of this shard", (long long)max_rows, (long long)h->info.nrows);
    if (c->device != h->device) return fail(CVR_ERR_INVALID, "communicator on device %d, matrix on device %d", c->device, h->device);
    const RcclApi *api = rccl_api();
    if (!api) return fail(CVR_ERR_NO_DEVICE, "RCCL not loaded");
    const hipStream_t    st = (hipStream_t)stream;
    const ncclDataType_t dt = h->vsz == 4 ? ncclFloat : ncclDouble;
    HIP_TRY(hipSetDevice(h->device));
    for (int b = 0; b < 2; b++)     // gathers an earlier overlapped call left on the communicator's stream
        if (c->pending[b]) { HIP_TRY(hipStreamWaitEvent(st, c->done[b], 0)); c->pending[b] = false; }
    if (!overlap) {                 // everything in order on the caller's stream: two enqueues per step, no events
        for (int k = 0; k < n; k++) {
            HIP_TRY(run_spmv(h, x_dev, y_dev[k & 1], st));
            RCCL_TRY(api, api->all_gather(y_dev[k & 1], yall_dev[k & 1], (size_t)max_rows, dt, c->comm, st));
        }
    } else {                        // the gather of step k (communicator's stream) overlaps the SpMV of step k + 1
        for (int k = 0; k < n; k++) {
            const int b = k & 1;
            if (c->pending[b]) HIP_TRY(hipStreamWaitEvent(st, c->done[b], 0));    // the gather that last read y_dev[b] / wrote yall_dev[b]
            HIP_TRY(run_spmv(h, x_dev, y_dev[b], st));
            HIP_TRY(hipEventRecord(c->ready[b], st));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ready[b], 0));
            RCCL_TRY(api, api->all_gather(y_dev[b], yall_dev[b], (size_t)max_rows, dt, c->comm, c->stream));
            HIP_TRY(hipEventRecord(c->done[b], c->stream));
            c->pending[b] = true;
        }
        for (int b = 0; b < 2; b++)
            if (c->pending[b]) { HIP_TRY(hipStreamWaitEvent(st, c->done[b], 0)); c->pending[b] = false; }
    }
    if (last_buf) *last_buf = n > 0 ? (n - 1) & 1 : 0;
    return CVR_OK;
}

// ---- the iterative caller: power iteration x <- A x / ||A x||, everything on the device ------------------------------
int cvr_power_iteration(cvr_handle *h, cvr_comm *c, const int64_t *bounds, int iters, void *x_dev, double *lambda,
                        double *seconds_per_iter, void *stream)
{
    if (!h || !x_dev || iters < 0) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_power_iteration before cvr_preprocess");
    const int     nparts = c ? c->nranks : 1;
    const int64_t n = h->info.ncols;                    // the whole (square) matrix has n rows and n columns
    if (c && !bounds) return fail(CVR_ERR_INVALID, "a communicator needs the row bounds of the shards");
    if (nparts > cvr::kIterMaxParts) return fail(CVR_ERR_INVALID, "more than %d shards", cvr::kIterMaxParts);
    cvr::IterBounds bd;
    int64_t         max_rows = 0;
    if (c) {
        if (c->device != h->device) return fail(CVR_ERR_INVALID, "communicator on device %d, matrix on device %d", c->device, h->device);
        if (bounds[0] != 0 || bounds[nparts] != n) return fail(CVR_ERR_INVALID, "bounds must run from 0 to ncols = %lld (square matrix)", (long long)n);
        for (int p = 0; p < nparts; p++) {
            if (bounds[p + 1] < bounds[p]) return fail(CVR_ERR_INVALID, "bounds decrease");
            max_rows = std::max(max_rows, bounds[p + 1] - bounds[p]);
        }
        for (int p = 0; p <= nparts; p++) bd.b[p] = bounds[p];
        if (bounds[c->rank + 1] - bounds[c->rank] != h->info.nrows) return fail(CVR_ERR_INVALID, "this rank's bounds do not match its %lld rows", (long long)h->info.nrows);
    } else if (h->info.nrows != n) {
        return fail(CVR_ERR_INVALID, "power iteration needs a square matrix (%lld x %lld)", (long long)h->info.nrows, (long long)n);
    }
    const RcclApi *api = c ? rccl_api() : nullptr;
    if (c && !api) return fail(CVR_ERR_NO_DEVICE, "RCCL not loaded");
    const hipStream_t st = (hipStream_t)stream;
    const bool        f32 = h->vsz == 4;
    HIP_TRY(hipSetDevice(h->device));

    // scratch: y_ext of this rank (room for the padded slice), the gathered padded y, the dense y, reduction cells
    struct Scratch {
        void *y = nullptr, *yall = nullptr, *dense = nullptr; double *partial = nullptr, *cells = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Scratch() { (void)hipFree(y); (void)hipFree(yall); (void)hipFree(dense); (void)hipFree(partial); (void)hipFree(cells);
                     if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } s;
    const size_t ny = (size_t)std::max<int64_t>(h->info.yext_elems, max_rows);
    HIP_TRY(hipMalloc(&s.y, h->vsz * std::max<size_t>(ny, 1)));
    HIP_TRY(hipMemsetAsync(s.y, 0, h->vsz * std::max<size_t>(ny, 1), st));
    if (c) {
        HIP_TRY(hipMalloc(&s.yall, h->vsz * std::max<size_t>((size_t)nparts * (size_t)max_rows, 1)));
        HIP_TRY(hipMalloc(&s.dense, h->vsz * std::max<size_t>((size_t)n, 1)));
    }
    const size_t npart = (size_t)std::max(cvr::dot_partials(), cvr::power_partials());
    HIP_TRY(hipMalloc(&s.partial, sizeof(double) * 2 * npart));      // two steps' partial sums, used in turn
    HIP_TRY(hipMalloc(&s.cells, sizeof(double) * 3));            // [0] = x . y, [1] = y . y, [2] = x . x of the last step
    HIP_TRY(hipMemsetAsync(s.cells, 0, sizeof(double) * 3, st));
    HIP_TRY(hipEventCreate(&s.e0));
    HIP_TRY(hipEventCreate(&s.e1));

    // x <- x / ||x||
    HIP_TRY(cvr::launch_dot(x_dev, x_dev, n, f32, s.partial, s.cells + 1, st));
    HIP_TRY(cvr::launch_scale(x_dev, x_dev, s.cells + 1, n, f32, st));
    HIP_TRY(hipEventRecord(s.e0, st));
    // The one-pass step scales x by the norm of the step BEFORE, so |x| swings up to ~lambda and y = A x up to ~lambda^2: fine in
    // fp64, but an fp32 handle whose dominant eigenvalue lies beyond ~1e15 (or below ~1e-15) would overflow (underflow) on the
    // way.  After the first step of such a handle the estimate |A x| / |x| is read back once; out of that range every further
    // step normalises exactly (two more passes over the vectors per step, |x| = 1 throughout).
    bool exact = false;
    for (int it = 0; it < iters; it++) {
        if (exact) {
            HIP_TRY(run_spmv(h, x_dev, s.y, st));
            const void *yfull = s.y;
            if (c) {
                RCCL_TRY(api, api->all_gather(s.y, s.yall, (size_t)max_rows, f32 ? ncclFloat : ncclDouble, c->comm, st));
                HIP_TRY(cvr::launch_unpad(s.dense, s.yall, bd, nparts, max_rows, f32, st));
                yfull = s.dense;
            }
            HIP_TRY(cvr::launch_dot(x_dev, yfull, n, f32, s.partial, s.cells + 0, st));
            HIP_TRY(cvr::launch_dot(x_dev, x_dev, n, f32, s.partial, s.cells + 2, st));
            HIP_TRY(cvr::launch_dot(yfull, yfull, n, f32, s.partial, s.cells + 1, st));
            HIP_TRY(cvr::launch_scale(x_dev, yfull, s.cells + 1, n, f32, st));
            continue;
        }
        HIP_TRY(run_spmv(h, x_dev, s.y, st));
        // the exchange step is on the critical path here: x of the next iteration is the gathered y (read through the shards'
        // bounds as it lies, padded)
        if (c) RCCL_TRY(api, api->all_gather(s.y, s.yall, (size_t)max_rows, f32 ? ncclFloat : ncclDouble, c->comm, st));
        // the step's three dot products and x <- y / ||y of the step before|| in one pass (cvr_iter.hip: power_step_kernel)
        HIP_TRY(cvr::launch_power_step(x_dev, c ? s.yall : s.y, n, f32, it > 0 ? s.partial + (size_t)((it - 1) & 1) * npart : nullptr,
                                       s.partial + (size_t)(it & 1) * npart, st, c ? &bd : nullptr, nparts, max_rows));
        if (f32 && it == 0 && iters > 1) {      // (one read-back per call, fp32 handles only)
            double part[3] = {0, 0, 0};
            HIP_TRY(cvr::launch_power_sums(s.partial, s.cells, st));
            HIP_TRY(hipMemcpyAsync(part, s.cells, sizeof(part), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const double est = part[2] > 0 ? sqrt(part[1] / part[2]) : 0.0;      // |A x| / |x|
            if (!(est > 1e-15 && est < 1e15)) {
                exact = true;                                                     // x holds y unscaled (prev was null): normalise it now
                const void *yfull = s.y;
                if (c) { HIP_TRY(cvr::launch_unpad(s.dense, s.yall, bd, nparts, max_rows, f32, st)); yfull = s.dense; }
                HIP_TRY(cvr::launch_scale(x_dev, yfull, s.cells + 1, n, f32, st));
                continue;
            }
        }
        if (it + 1 == iters) {       // the last iterate leaves normalised exactly: x <- y / ||y||
            const void *yfull = s.y;
            if (c) { HIP_TRY(cvr::launch_unpad(s.dense, s.yall, bd, nparts, max_rows, f32, st)); yfull = s.dense; }
            HIP_TRY(cvr::launch_power_sums(s.partial + (size_t)(it & 1) * npart, s.cells, st));
            HIP_TRY(cvr::launch_scale(x_dev, yfull, s.cells + 1, n, f32, st));
        }
    }
    HIP_TRY(hipEventRecord(s.e1, st));
    double cells[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(cells, s.cells, sizeof(cells), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s.e0, s.e1));
    if (lambda) *lambda = iters > 0 && cells[2] > 0 ? cells[0] / cells[2] : 0.0;      // Rayleigh quotient of the last step's x
    if (seconds_per_iter) *seconds_per_iter = iters > 0 ? (double)ms * 1e-3 / iters : 0.0;
    return CVR_OK;
}

int cvr_spmv_bench(cvr_handle *h, int warmup, int iters, double *mean_s)
{
    if (!h || iters < 1) return fail(CVR_ERR_INVALID, "bad argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    HIP_TRY(hipSetDevice(h->device));
    if (h->events.size() < 2) {
        h->events.resize(2);
        HIP_TRY(hipEventCreate(&h->events[0]));
        HIP_TRY(hipEventCreate(&h->events[1]));
    }
    for (int i = 0; i < warmup; i++) HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));
    HIP_TRY(hipEventRecord(h->events[0], h->stream));
    for (int i = 0; i < iters; i++) HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));
    HIP_TRY(hipEventRecord(h->events[1], h->stream));
    HIP_TRY(hipEventSynchronize(h->events[1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->events[0], h->events[1]));
    if (mean_s) *mean_s = ms * 1e-3 / iters;
    return CVR_OK;
}

int cvr_spmv(cvr_handle *h, const void *x_host, void *y_host, int iters, cvr_timing *tm)
{
    if (!h || !x_host || !y_host) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    if (iters < 1) iters = 1;
    Range range("cvr_spmv (h2d x, timed launches, d2h y)");
    HIP_TRY(hipSetDevice(h->device));
    const size_t need = (size_t)iters + 1;
    while (h->events.size() < need) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->events.push_back(e);
    }
    double t0 = now_s();
    if (h->info.ncols) HIP_TRY(hipMemcpyAsync(h->d_x, x_host, h->vsz * (size_t)h->info.ncols, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double h2d = now_s() - t0;
    HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));   // warm-up, untimed
    HIP_TRY(hipEventRecord(h->events[0], h->stream));
    for (int i = 0; i < iters; i++) {
        HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));
        HIP_TRY(hipEventRecord(h->events[(size_t)i + 1], h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    t0 = now_s();
    if (h->info.nrows) HIP_TRY(hipMemcpyAsync(y_host, h->d_y, h->vsz * (size_t)h->info.nrows, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double d2h = now_s() - t0;
    if (tm) {
        memset(tm, 0, sizeof(*tm));
        tm->iters = iters; tm->h2d_s = h2d; tm->d2h_s = d2h;
        double sum = 0, mn = 1e30, mx = 0;
        for (int i = 0; i < iters; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, h->events[(size_t)i], h->events[(size_t)i + 1]));
            const double s = ms * 1e-3;
            sum += s; mn = std::min(mn, s); mx = std::max(mx, s);
        }
        float tot = 0;
        HIP_TRY(hipEventElapsedTime(&tot, h->events[0], h->events[(size_t)iters]));
        tm->mean_s = sum / iters; tm->min_s = mn; tm->max_s = mx; tm->total_s = tot * 1e-3;
    }
    return CVR_OK;
}

int cvr_auto_panels(const cvr_csr_view *csr, double *l2_miss_estimate_out)
{
    if (!csr) return fail(CVR_ERR_INVALID, "null argument");
    if (csr->arrays_on_device) return fail(CVR_ERR_INVALID, "cvr_auto_panels reads host arrays");
    if (csr->nrows > 0 && (!csr->row_ptr || (csr->row_ptr[csr->nrows] > 0 && !csr->col_idx))) return fail(CVR_ERR_INVALID, "null argument");
    return auto_panels(*csr, l2_miss_estimate_out);
}

static int tune_impl(const cvr_csr_view *csr, const cvr_options *opt_in, bool full_layout, cvr_options *best_out, double *best_spmv_s, double *tuning_s)
{
    cvr_options opt;
    if (opt_in) opt = *opt_in; else cvr_default_options(&opt);
    const double t0 = now_s();
    // host arrays go to the device once; every candidate is then built from the device copy (device-to-device, no PCIe)
    cvr_csr_view view = *csr;
    struct Staged { void *rp = nullptr, *ci = nullptr, *va = nullptr; ~Staged() { (void)hipFree(rp); (void)hipFree(ci); (void)hipFree(va); } } staged;
    if (!csr->arrays_on_device && cvr_device_count() > 0 && csr->nrows > 0 && csr->row_ptr && csr->row_ptr[csr->nrows] > 0 && csr->col_idx && csr->vals) {
        int rc = check_csr(csr);
        if (rc) return rc;
        const size_t nz = (size_t)csr->row_ptr[csr->nrows], vs = csr->is_f32 ? 4 : 8;
        HIP_TRY(hipSetDevice(opt.device));
        HIP_TRY(hipMalloc(&staged.rp, sizeof(int64_t) * ((size_t)csr->nrows + 1)));
        HIP_TRY(hipMalloc(&staged.ci, sizeof(int32_t) * nz));
        HIP_TRY(hipMalloc(&staged.va, vs * nz));
        HIP_TRY(hipMemcpy(staged.rp, csr->row_ptr, sizeof(int64_t) * ((size_t)csr->nrows + 1), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(staged.ci, csr->col_idx, sizeof(int32_t) * nz, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(staged.va, csr->vals, vs * nz, hipMemcpyHostToDevice));
        view.row_ptr = static_cast<const int64_t *>(staged.rp);
        view.col_idx = static_cast<const int32_t *>(staged.ci);
        view.vals = staged.va;
        view.arrays_on_device = 1;
        if (opt.col_panels < 0) opt.col_panels = auto_panels(*csr, nullptr);      // decided once, on the host arrays
    }
    // every candidate is the real thing: plan, convert, timed launches
    auto measure = [&](const cvr_options &o, double *t_out) -> int {
        cvr_handle *h = nullptr;
        int         rc = cvr_create(&h, &view, &o);
        double      t = 0;
        if (rc == CVR_OK) rc = cvr_preprocess(h, 0, nullptr);
        if (rc == CVR_OK) rc = cvr_spmv_bench(h, 5, 10, &t);                                   // settle clocks and caches
        if (rc == CVR_OK) rc = cvr_spmv_bench(h, 0, t > 0 ? std::max(20, std::min(200, (int)(1.5e-3 / t))) : 20, &t);
        cvr_destroy(h);
        *t_out = t;
        return rc;
    };
    cvr_options best = opt;
    double      best_t = 0;
    bool        have = false;
    // (1) one chunk per workgroup, S = 8 .. 64
    // (waves_per_block stays 0 = default: an explicit 1 would switch the automatic hub table off, cvr_layout: choose_hubs, and a
    // tuned handle of a power-law shard would lose its tables)
    opt.waves_per_block = 0; opt.x_window = 0; opt.col_phases = 1;
    for (int32_t S = 8; S <= 64; S += 4) {
        opt.steps_per_chunk = S;
        double t = 0;
        const int rc = measure(opt, &t);
        if (rc != CVR_OK) return rc;
        if (!have || t < best_t) { have = true; best = opt; best_t = t; }
    }
    // (2) the resident layout (several chunks per workgroup, one workgroup per CU, all at once) where the matrix is small
    // enough: 64-KiB window of x or none, column phases or none; skipped with column panels and without row pointers here
    if (full_layout && opt.col_panels <= 1 && csr->nrows > 0 && !csr->arrays_on_device) {
        const int64_t vs = csr->is_f32 ? 4 : 8;
        const double  slots = ((double)(csr->row_ptr[csr->nrows] - csr->row_ptr[0]) + (double)csr->nrows / 4) * 1.006;
        const double  xbytes = (double)csr->ncols * vs;
        const int     P = (int)std::min(32.0, std::max(2.0, std::floor(xbytes / 600e3 + 0.5)));
        for (int w : {8, 7, 4}) {
            int S = (int)std::ceil(slots / (64.0 * w * 252.0) / 4.0) * 4;
            if (S < 8) S = 8;
            if (S > 128) continue;
            for (int win : {(int)(65536 / vs), 0})
                for (int ph : {P, 1}) {
                    if (xbytes <= 2.5e6 && ph > 1) continue;
                    cvr_options o = opt;
                    o.waves_per_block = w; o.steps_per_chunk = S; o.x_window = win; o.col_phases = ph;
                    double t = 0;
                    const int rc = measure(o, &t);
                    if (rc == CVR_ERR_INVALID) { (void)hipGetLastError(); continue; }     // e.g. unsorted rows with phases: not a candidate
                    if (rc != CVR_OK) return rc;
                    if (t < best_t) { best = o; best_t = t; }
                }
        }
    }
    *best_out = best;
    if (best_spmv_s) *best_spmv_s = best_t;
    if (tuning_s) *tuning_s = now_s() - t0;
    return CVR_OK;
}

int cvr_tune(const cvr_csr_view *csr, const cvr_options *opt_in, cvr_options *best, double *best_spmv_s, double *tuning_s)
{
    if (!csr || !best) return fail(CVR_ERR_INVALID, "null argument");
    return tune_impl(csr, opt_in, true, best, best_spmv_s, tuning_s);
}

int cvr_tune_steps(const cvr_csr_view *csr, const cvr_options *opt_in, int32_t *best_steps, double *best_spmv_s, double *tuning_s)
{
    if (!csr || !best_steps) return fail(CVR_ERR_INVALID, "null argument");
    cvr_options best;
    const int   rc = tune_impl(csr, opt_in, false, &best, best_spmv_s, tuning_s);
    if (rc == CVR_OK) *best_steps = best.steps_per_chunk;
    return rc;
}

int cvr_device_copy_bench(int device, int64_t bytes, int iters, double *gbs)
{
    if (bytes < 16 || iters < 1) return fail(CVR_ERR_INVALID, "bad argument");
    if (device < 0 || device >= cvr_device_count()) return fail(CVR_ERR_NO_DEVICE, "device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    void       *a = nullptr, *b = nullptr;
    hipStream_t st;
    hipEvent_t  e0, e1;
    HIP_TRY(hipMalloc(&a, (size_t)bytes));
    HIP_TRY(hipMalloc(&b, (size_t)bytes));
    HIP_TRY(hipStreamCreate(&st));
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipMemsetAsync(a, 1, (size_t)bytes, st));
    for (int i = 0; i < 3; i++) HIP_TRY(cvr::launch_copy(a, b, (size_t)bytes, st));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) HIP_TRY(cvr::launch_copy(a, b, (size_t)bytes, st));
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (gbs) *gbs = 2.0 * (double)(bytes / 16 * 16) * iters / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st); (void)hipFree(a); (void)hipFree(b);
    return CVR_OK;
}

int cvr_export_image(cvr_handle *h, void *stream_image, uint32_t *desc, uint8_t *target, int64_t *shared)
{
    if (!h) return fail(CVR_ERR_INVALID, "handle is null");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_export_image before cvr_preprocess");
    if (h->paneled()) return fail(CVR_ERR_STATE, "cvr_export_image exports one image: create the handle with col_panels = 1");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const Part  &p = h->parts[0];
    const size_t nc = (size_t)p.nchunks;
    if (stream_image && p.stream_bytes) HIP_TRY(hipMemcpy(stream_image, p.img.stream, p.stream_bytes, hipMemcpyDeviceToHost));
    if (desc && nc) HIP_TRY(hipMemcpy(desc, p.img.desc, 16 * nc, hipMemcpyDeviceToHost));
    if (target && nc) HIP_TRY(hipMemcpy(target, p.img.target, 64 * nc, hipMemcpyDeviceToHost));
    if (shared && p.nshared) HIP_TRY(hipMemcpy(shared, p.img.shared, 24 * (size_t)p.nshared, hipMemcpyDeviceToHost));
    return CVR_OK;
}

}  // extern "C"

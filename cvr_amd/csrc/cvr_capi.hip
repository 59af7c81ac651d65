// cvr_capi.hip -- the C ABI of include/cvr_amd.h: handle life cycle, device memory, timing.
// Host orchestration that the reference keeps in main() (allocation block spmv.cpp:1777-1829, calls at
// spmv.cpp:1857 and 1882) lives behind the handle here; the caller keeps only CSR, x and y.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
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
        static push_t p = nullptr;
        static pop_t  q = nullptr;
        static bool   tried = false;
        if (!tried) {
            tried = true;
            const char *e = getenv("CVR_ROCTX");
            if (e && atoi(e)) {
                void *lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
                if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (lib) { p = (push_t)dlsym(lib, "roctxRangePushA"); q = (pop_t)dlsym(lib, "roctxRangePop"); }
            }
        }
        push = p; pop = q;
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

struct cvr_handle {
    int              device = 0;
    hipStream_t      stream = nullptr;
    bool             converted = false;
    cvr_info         info{};
    cvr::DeviceImage img{};
    // device CSR + plan (dropped after conversion unless keep_csr)
    int64_t  *d_rp = nullptr;
    int32_t  *d_ci = nullptr;
    void     *d_va = nullptr;
    int64_t  *d_nzb = nullptr;
    uint32_t *d_pad = nullptr;
    uint32_t *d_err = nullptr;
    void     *d_x = nullptr;   // x_ext: ncols + 1
    void     *d_y = nullptr;   // y_ext
    size_t    vsz = 8;
    size_t    stream_bytes = 0;
    std::vector<hipEvent_t> events;

    void release_csr()
    {
        if (d_rp) (void)hipFree(d_rp);
        if (d_ci) (void)hipFree(d_ci);
        if (d_va) (void)hipFree(d_va);
        if (d_nzb) (void)hipFree(d_nzb);
        if (d_pad) (void)hipFree(d_pad);
        d_rp = nullptr; d_ci = nullptr; d_va = nullptr; d_nzb = nullptr; d_pad = nullptr;
    }
};

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
}

int cvr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int64_t cvr_plan_bound(int64_t nrows, int64_t nnz, int32_t S) { return cvr::plan_bound(nrows, nnz, S); }

static int check_csr(const cvr_csr_view *c)
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
    for (int64_t j = c->row_ptr[0]; j < nnz; j++)
        if (c->col_idx[j] < 0 || c->col_idx[j] >= c->ncols)
            return fail(CVR_ERR_INVALID, "col_idx[%lld] = %d outside [0, %lld)", (long long)j, c->col_idx[j], (long long)c->ncols);
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

static int pick_steps(int64_t nslots_est)
{
    // the longest lane streams (fewest per-chunk prologues, fewest cut rows) that still leave >= 8 wavefronts
    // for every one of the 256 CUs: S = 32 for web-Google (2.6 k chunks), 128 for soc-LiveJournal1
    const int64_t want_chunks = 256 * 8;
    int           S = 128;
    while (S > 16 && nslots_est / (64 * (int64_t)S) < want_chunks) S /= 2;
    return S;
}

int cvr_create(cvr_handle **out, const cvr_csr_view *csr, const cvr_options *opt_in)
{
    if (!out) return fail(CVR_ERR_INVALID, "out is null");
    *out = nullptr;
    Range range("cvr_create (validate, plan, upload)");
    int rc = check_csr(csr);
    if (rc) return rc;
    cvr_options opt;
    if (opt_in) opt = *opt_in; else cvr_default_options(&opt);
    const int ndev = cvr_device_count();
    if (ndev <= 0) return fail(CVR_ERR_NO_DEVICE, "no HIP device visible: libcvr_amd has no CPU fallback");
    if (opt.device < 0 || opt.device >= ndev) return fail(CVR_ERR_NO_DEVICE, "device %d out of range [0, %d)", opt.device, ndev);

    const int64_t nrows = csr->nrows, ncols = csr->ncols;
    const int64_t nz0 = nrows ? csr->row_ptr[0] : 0, nz1 = nrows ? csr->row_ptr[nrows] : 0;
    int S = opt.steps_per_chunk;
    if (S == 0) S = pick_steps(nz1 - nz0 + nrows / 4);
    if (S < 4 || S % 4 || S > 4096) return fail(CVR_ERR_INVALID, "steps_per_chunk must be a multiple of 4 in [4, 4096]");

    cvr_handle *h = new (std::nothrow) cvr_handle;
    if (!h) return fail(CVR_ERR_NOMEM, "out of host memory");
    h->device = opt.device;
    h->vsz = csr->is_f32 ? 4 : 8;

    const double    t0 = now_s();
    const cvr::Plan plan = cvr::plan_chunks(nrows, csr->row_ptr, S, opt.split_threshold);
    const int64_t   nchunks = (int64_t)plan.chunks.size();
    const int64_t   yext = nrows + 1 + 2 * nchunks;
    if (yext >= (int64_t)0xffffffffu || nchunks >= (int64_t)0x7fffffff) { delete h; return fail(CVR_ERR_INVALID, "matrix too large for 32-bit row ordinals on one GPU"); }
    std::vector<uint32_t> desc((size_t)nchunks * 4), pad((size_t)nchunks);
    std::vector<int64_t>  nzb((size_t)nchunks + 1);
    for (int64_t k = 0; k < nchunks; k++) {
        const cvr::Chunk &c = plan.chunks[(size_t)k];
        const uint32_t    rf = (uint32_t)c.row_first, ns = (uint32_t)c.nseg;
        desc[4 * k + 0] = rf;
        desc[4 * k + 1] = ns;
        // where segment q writes: a row begun earlier -> carry_head(k); a row continued later -> carry_tail(k);
        // the pad segment -> dump; else its row.  head_dest / last_dest are that rule at q = 0 and q = nseg-1.
        auto dest = [&](int64_t q) -> uint32_t {
            if (q >= c.nrows_in) return (uint32_t)nrows;
            if (q == 0 && c.head_shared) return (uint32_t)(nrows + 1 + 2 * k);
            if (q == c.nrows_in - 1 && c.tail_shared) return (uint32_t)(nrows + 1 + 2 * k + 1);
            return (uint32_t)(c.row_first + q);
        };
        desc[4 * k + 2] = dest(0);
        desc[4 * k + 3] = dest(c.nseg - 1);
        pad[(size_t)k] = (uint32_t)c.pad_cnt;
        nzb[(size_t)k] = c.nz_begin;
    }
    nzb[(size_t)nchunks] = plan.nz_end;
    const double t1 = now_s();

    cvr_info &in = h->info;
    in.nrows = nrows; in.ncols = ncols; in.nnz = nz1 - nz0; in.is_f32 = csr->is_f32 ? 1 : 0; in.steps_per_chunk = S;
    in.nchunks = nchunks; in.nslots = nchunks * 64 * S; in.nshared = (int64_t)plan.shared.size();
    in.yext_elems = yext; in.x_elems = ncols + 1; in.plan_s = t1 - t0;
    const int G = S / 4;
    h->stream_bytes = (size_t)nchunks * G * cvr::group_bytes(csr->is_f32 != 0);
    in.image_bytes = (int64_t)(h->stream_bytes + (size_t)nchunks * (16 + 64) + plan.shared.size() * 24);

    cvr::DeviceImage &img = h->img;
    img.S = S; img.G = G; img.f32 = csr->is_f32 != 0; img.nchunks = (uint32_t)nchunks; img.nrows = (uint32_t)nrows;
    img.pad_col = (uint32_t)ncols; img.nshared = (uint32_t)plan.shared.size();
    img.xcd_swizzle = opt.xcd_swizzle != 0;
    img.stream_policy = opt.stream_policy > 0 ? opt.stream_policy : 0;
    img.gather_policy = opt.gather_policy > 0 ? opt.gather_policy : 0;
    img.depth = opt.gather_depth == 2 ? 2 : 1;
    // LDS window of x per workgroup: off by default.  Measured on MI355X (profiles/r01_lds_window_sweep.log): the
    // gathers it absorbs are the cheap ones (L1/L2 hits near the diagonal); the kernel's time is set by the L2
    // misses of the scattered columns, which a contiguous window cannot hold, and the LDS it takes costs occupancy.
    int64_t win = opt.x_window < 0 ? 0 : opt.x_window;
    if (win > ncols + 1) win = ncols + 1;
    if (win > 16384) win = 16384;                     // 128 KiB of fp64 + the steal slots stay under 160 KiB
    img.win_elems = (uint32_t)win;
    if (opt.debug_col_mask) img.col_mask = (uint32_t)opt.debug_col_mask & cvr::kColMask;   // profiling knob (tools/sweep.py --colmask)

#define HIP_TRY_H(expr)                                                                                     \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);          \
            cvr_destroy(h);                                                                                 \
            return CVR_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)
    HIP_TRY_H(hipSetDevice(h->device));
    HIP_TRY_H(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const size_t nnz_span = (size_t)nz1;   // arrays are indexed literally from 0
    HIP_TRY_H(hipMalloc(&h->d_rp, sizeof(int64_t) * ((size_t)nrows + 1)));
    HIP_TRY_H(hipMalloc(&h->d_ci, sizeof(int32_t) * std::max<size_t>(nnz_span, 1)));
    HIP_TRY_H(hipMalloc(&h->d_va, h->vsz * std::max<size_t>(nnz_span, 1)));
    HIP_TRY_H(hipMalloc(&h->d_nzb, sizeof(int64_t) * ((size_t)nchunks + 1)));
    HIP_TRY_H(hipMalloc(&h->d_pad, sizeof(uint32_t) * std::max<size_t>((size_t)nchunks, 1)));
    HIP_TRY_H(hipMalloc(&h->d_err, sizeof(uint32_t)));
    if (getenv("CVR_STREAM_UNCACHED"))   // experiment: matrix image in uncached (MTYPE UC) memory, so that it cannot displace x in L2
        HIP_TRY_H(hipExtMallocWithFlags((void **)&img.stream, std::max<size_t>(h->stream_bytes, 16), hipDeviceMallocUncached));
    else
        HIP_TRY_H(hipMalloc(&img.stream, std::max<size_t>(h->stream_bytes, 16)));
    HIP_TRY_H(hipMalloc(&img.desc, 16 * std::max<size_t>((size_t)nchunks, 1)));
    HIP_TRY_H(hipMalloc(&img.target, 64 * std::max<size_t>((size_t)nchunks, 1)));
    HIP_TRY_H(hipMalloc(&img.shared, 24 * std::max<size_t>(plan.shared.size(), 1)));
    HIP_TRY_H(hipMalloc(&img.win_base, sizeof(uint32_t) * ((size_t)nchunks / cvr::kWavesPerBlock + 1)));
    HIP_TRY_H(hipMemsetAsync(img.win_base, 0, sizeof(uint32_t) * ((size_t)nchunks / cvr::kWavesPerBlock + 1), h->stream));
    HIP_TRY_H(hipMalloc(&h->d_x, h->vsz * (size_t)in.x_elems));
    HIP_TRY_H(hipMalloc(&h->d_y, h->vsz * (size_t)in.yext_elems));
    HIP_TRY_H(hipMemsetAsync(h->d_x, 0, h->vsz * (size_t)in.x_elems, h->stream));
    HIP_TRY_H(hipMemsetAsync(h->d_y, 0, h->vsz * (size_t)in.yext_elems, h->stream));
    HIP_TRY_H(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
    const double t2 = now_s();
    if (nrows > 0) HIP_TRY_H(hipMemcpyAsync(h->d_rp, csr->row_ptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice, h->stream));
    if (nnz_span) {
        HIP_TRY_H(hipMemcpyAsync(h->d_ci, csr->col_idx, sizeof(int32_t) * nnz_span, hipMemcpyHostToDevice, h->stream));
        HIP_TRY_H(hipMemcpyAsync(h->d_va, csr->vals, h->vsz * nnz_span, hipMemcpyHostToDevice, h->stream));
    }
    if (nchunks) {
        HIP_TRY_H(hipMemcpyAsync(h->d_nzb, nzb.data(), sizeof(int64_t) * nzb.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY_H(hipMemcpyAsync(h->d_pad, pad.data(), sizeof(uint32_t) * pad.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY_H(hipMemcpyAsync(img.desc, desc.data(), sizeof(uint32_t) * desc.size(), hipMemcpyHostToDevice, h->stream));
    }
    if (!plan.shared.empty())
        HIP_TRY_H(hipMemcpyAsync(img.shared, plan.shared.data(), sizeof(cvr::Shared) * plan.shared.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY_H(hipStreamSynchronize(h->stream));   // the caller may free its CSR when this returns
    in.upload_s = now_s() - t2;
#undef HIP_TRY_H
    *out = h;
    return CVR_OK;
}

int cvr_preprocess(cvr_handle *h, int keep_csr, double *seconds)
{
    if (!h) return fail(CVR_ERR_INVALID, "handle is null");
    if (!h->d_rp) return fail(CVR_ERR_STATE, "the device CSR was already released: cvr_preprocess runs once unless keep_csr was set");
    Range range("cvr_preprocess (CSR -> CVR64)");
    HIP_TRY(hipSetDevice(h->device));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
    cvr::DeviceCsr csr;
    csr.row_ptr = h->d_rp; csr.col_idx = h->d_ci; csr.vals = h->d_va; csr.nz_begin = h->d_nzb; csr.pad_cnt = h->d_pad;
    HIP_TRY(hipEventRecord(e0, h->stream));
    HIP_TRY(cvr::launch_convert(h->img, csr, h->d_err, h->stream));
    HIP_TRY(cvr::launch_window(h->img, csr, h->stream));
    HIP_TRY(hipEventRecord(e1, h->stream));
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, h->d_err, sizeof(err), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    h->info.convert_s = ms * 1e-3;
    if (seconds) *seconds = ms * 1e-3;
    if (err) return fail(CVR_ERR_INTERNAL, "device converter self-check failed (flags 0x%x)", err);
    h->converted = true;
    if (!keep_csr) h->release_csr();
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
    h->release_csr();
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    if (h->d_err) (void)hipFree(h->d_err);
    if (h->img.stream) (void)hipFree(h->img.stream);
    if (h->img.desc) (void)hipFree(h->img.desc);
    if (h->img.target) (void)hipFree(h->img.target);
    if (h->img.shared) (void)hipFree(h->img.shared);
    if (h->img.win_base) (void)hipFree(h->img.win_base);
    if (h->d_x) (void)hipFree(h->d_x);
    if (h->d_y) (void)hipFree(h->d_y);
    if (h->stream) (void)hipStreamDestroy(h->stream);
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
    HIP_TRY(cvr::launch_spmv(h->img, x_dev, y_dev, (hipStream_t)stream));
    return CVR_OK;
}

int cvr_spmv_device_repeat(cvr_handle *h, const void *x_dev, void *y_dev, void *stream, int n)
{
    if (!h || !x_dev || !y_dev) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    const hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; i++) HIP_TRY(cvr::launch_spmv(h->img, x_dev, y_dev, st));
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
    for (int i = 0; i < warmup; i++) HIP_TRY(cvr::launch_spmv(h->img, h->d_x, h->d_y, h->stream));
    HIP_TRY(hipEventRecord(h->events[0], h->stream));
    for (int i = 0; i < iters; i++) HIP_TRY(cvr::launch_spmv(h->img, h->d_x, h->d_y, h->stream));
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
    HIP_TRY(cvr::launch_spmv(h->img, h->d_x, h->d_y, h->stream));   // warm-up, untimed
    HIP_TRY(hipEventRecord(h->events[0], h->stream));
    for (int i = 0; i < iters; i++) {
        HIP_TRY(cvr::launch_spmv(h->img, h->d_x, h->d_y, h->stream));
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
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t nc = (size_t)h->info.nchunks;
    if (stream_image && h->stream_bytes) HIP_TRY(hipMemcpy(stream_image, h->img.stream, h->stream_bytes, hipMemcpyDeviceToHost));
    if (desc && nc) HIP_TRY(hipMemcpy(desc, h->img.desc, 16 * nc, hipMemcpyDeviceToHost));
    if (target && nc) HIP_TRY(hipMemcpy(target, h->img.target, 64 * nc, hipMemcpyDeviceToHost));
    if (shared && h->info.nshared) HIP_TRY(hipMemcpy(shared, h->img.shared, 24 * (size_t)h->info.nshared, hipMemcpyDeviceToHost));
    return CVR_OK;
}

}  // extern "C"

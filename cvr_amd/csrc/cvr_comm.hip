// cvr_comm.hip -- the exchange step of the row-sharded SpMV (RCCL over xGMI, one process per GPU: cvr_comm_*, cvr_spmv_gather_repeat)
// and the iterative caller (cvr_power_iteration).
#include "cvr_internal.h"

using namespace cvrh;

namespace cvrh {

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
            a.comm_init_all = (decltype(a.comm_init_all))dlsym(a.lib, "ncclCommInitAll");
            a.group_start = (decltype(a.group_start))dlsym(a.lib, "ncclGroupStart");
            a.group_end = (decltype(a.group_end))dlsym(a.lib, "ncclGroupEnd");
            a.comm_count = (decltype(a.comm_count))dlsym(a.lib, "ncclCommCount");
            a.comm_user_rank = (decltype(a.comm_user_rank))dlsym(a.lib, "ncclCommUserRank");
            a.get_version = (decltype(a.get_version))dlsym(a.lib, "ncclGetVersion");
            if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather || !a.error_string || !a.comm_init_all || !a.group_start || !a.group_end) a.lib = nullptr;
        }
        return a;
    }();
    return api.lib ? &api : nullptr;
}


}  // namespace cvrh

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

int cvr_comm_info(cvr_comm *c, int *nranks, int *rank, int *rccl_version)
{
    if (!c) return fail(CVR_ERR_INVALID, "communicator is null");
    const RcclApi *api = rccl_api();
    if (!api || !c->comm) return fail(CVR_ERR_STATE, "no RCCL communicator behind this handle");
    int n = -1, r = -1, v = -1;
    if (api->comm_count) RCCL_TRY(api, api->comm_count(c->comm, &n));
    if (api->comm_user_rank) RCCL_TRY(api, api->comm_user_rank(c->comm, &r));
    if (api->get_version) RCCL_TRY(api, api->get_version(&v));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (rccl_version) *rccl_version = v;
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
    cvr::debug_refresh();
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
    // One GPU, an image with column phases and no rows cut over chunks (the resident layout of a web-graph-sized matrix): the step's dot
    // products and the next iterate come out of the SpMV kernel's write-out (cvr_kernels.h: IterEpilogue) -- one launch per iteration,
    // x alternating between the caller's buffer and one of ours.
    bool         fused = !c && !h->paneled() && cvr::iter_epilogue_ok(h->parts[0].img) && h->parts[0].img.hub_n == 0 && !cvr::debug_env("iter_unfused");
    void        *xalt = nullptr;
    struct AltGuard { void *&p; ~AltGuard() { (void)hipFree(p); } } alt_guard{xalt};
    const size_t nsets = (size_t)cvr::power_partials() / 3;
    if (fused) {
        HIP_TRY(hipMalloc(&xalt, h->vsz * (size_t)h->info.x_elems));
        HIP_TRY(hipMemsetAsync(xalt, 0, h->vsz * (size_t)h->info.x_elems, st));
        HIP_TRY(hipMemsetAsync(s.partial, 0, sizeof(double) * 2 * npart, st));      // (the kernel writes one entry per workgroup and set: the rest stays zero)
    }
    void *cur = x_dev, *nxt = xalt;
    for (int it = 0; it < iters; it++) {
        if (fused) {
            cvr::IterEpilogue epi;
            epi.xnext = nxt; epi.prev = it > 0 ? s.partial + (size_t)((it - 1) & 1) * npart : nullptr; epi.out = s.partial + (size_t)(it & 1) * npart; epi.nsets = (uint32_t)nsets;
            HIP_TRY(cvr::launch_spmv(h->parts[0].img, cur, s.y, st, true, nullptr, 0, 1, &epi));
            std::swap(cur, nxt);
            if (f32 && it == 0 && iters > 1) {      // the fp32 range check of the unfused loop below
                double part[3] = {0, 0, 0};
                HIP_TRY(cvr::launch_power_sums(s.partial, s.cells, st));
                HIP_TRY(hipMemcpyAsync(part, s.cells, sizeof(part), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                const double est = part[2] > 0 ? sqrt(part[1] / part[2]) : 0.0;
                if (!(est > 1e-15 && est < 1e15)) {
                    exact = true; fused = false;
                    HIP_TRY(cvr::launch_scale(x_dev, s.y, s.cells + 1, n, f32, st));
                    continue;
                }
            }
            if (it + 1 == iters) {       // the last iterate leaves normalised exactly, in the caller's buffer: x <- y / ||y||
                HIP_TRY(cvr::launch_power_sums(s.partial + (size_t)(it & 1) * npart, s.cells, st));
                HIP_TRY(cvr::launch_scale(x_dev, s.y, s.cells + 1, n, f32, st));
            }
            continue;
        }
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

}  // extern "C"

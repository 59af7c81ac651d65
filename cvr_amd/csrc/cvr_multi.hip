// cvr_multi.hip -- one call = all GPUs of the process: the multi-device handle of include/cvr_amd.h (cvr_create_multi,
// cvr_preprocess_multi, cvr_spmv_multi) and the one row-partition rule (cvr_row_partition).
//
// The reference's two entry points each drive ALL threads of the machine (pre_processing spmv.cpp:1857 -> `omp parallel
// num_threads(Nthrds)` at :577; spmv_compute_kernel :1882 -> :1034), with one y shared in host memory and atomics where two
// threads meet in a row (spmv.cpp:1280-1282, 1640-1649).  Here one host thread drives one shard handle per GPU on that GPU's
// own stream: rows are cut at row boundaries with balanced non-zeros (no row spans devices: nothing to reduce), x is
// replicated, and every step ends with the all-gather of the padded y slices (ncclAllGather inside a group call; between
// entries that name the same device, device-to-device copies).
#include "cvr_internal.h"

using namespace cvrh;

struct cvr_multi {
    int                       G = 0;
    bool                      f32 = false, use_rccl = false, converted = false;
    int64_t                   nrows = 0, ncols = 0, max_rows = 0;
    std::vector<int>          dev;
    std::vector<int64_t>      bounds;          // [G + 1]
    std::vector<cvr_handle *> H;
    std::vector<void *>       dx, dy, dall;    // per device: replicated x_ext, the shard's y_ext (at least max_rows), the gathered y (G * max_rows)
    std::vector<hipStream_t>  st;
    std::vector<ncclComm_t>   comm;
    std::vector<std::vector<hipEvent_t>> ev;   // per device: iters + 1 events of a timed loop
    size_t                    vsz = 8;
};

namespace {

int multi_sync(cvr_multi *m)
{
    for (int g = 0; g < m->G; g++) { HIP_TRY(hipSetDevice(m->dev[(size_t)g])); HIP_TRY(hipStreamSynchronize(m->st[(size_t)g])); }
    return CVR_OK;
}

// one step on every device: the shards' SpMVs, then (gather) the exchange of the y slices
int multi_step(cvr_multi *m, bool gather)
{
    const RcclApi *api = m->use_rccl ? rccl_api() : nullptr;
    for (int g = 0; g < m->G; g++) {
        HIP_TRY(hipSetDevice(m->dev[(size_t)g]));
        HIP_TRY(run_spmv(m->H[(size_t)g], m->dx[(size_t)g], m->dy[(size_t)g], m->st[(size_t)g]));
    }
    if (!gather || m->G == 1) return CVR_OK;
    if (api) {
        RCCL_TRY(api, api->group_start());
        for (int g = 0; g < m->G; g++)
            RCCL_TRY(api, api->all_gather(m->dy[(size_t)g], m->dall[(size_t)g], (size_t)m->max_rows, m->f32 ? ncclFloat : ncclDouble, m->comm[(size_t)g], m->st[(size_t)g]));
        RCCL_TRY(api, api->group_end());
    } else {
        // the same delivery by copies: every "rank" receives every slice; the copy of slice src is ordered behind src's SpMV on
        // src's stream (entries that name the same device share it; distinct devices without RCCL are not produced by cvr_create_multi)
        for (int g = 0; g < m->G; g++)
            for (int src = 0; src < m->G; src++) {
                HIP_TRY(hipSetDevice(m->dev[(size_t)src]));
                HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(m->dall[(size_t)g]) + (size_t)src * (size_t)m->max_rows * m->vsz, m->dy[(size_t)src],
                                       m->vsz * (size_t)m->max_rows, hipMemcpyDeviceToDevice, m->st[(size_t)src]));
            }
    }
    return CVR_OK;
}

// `iters` timed steps: per step the slowest device counts; mean / min / median / max over the steps
int multi_timed(cvr_multi *m, bool gather, int iters, double *mean, double *mn, double *med, double *mx, double *total)
{
    for (int w = 0; w < 3; w++) { const int rc = multi_step(m, gather); if (rc) return rc; }
    int rc = multi_sync(m);
    if (rc) return rc;
    for (int g = 0; g < m->G; g++) {
        HIP_TRY(hipSetDevice(m->dev[(size_t)g]));
        while (m->ev[(size_t)g].size() < (size_t)iters + 1) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); m->ev[(size_t)g].push_back(e); }
        HIP_TRY(hipEventRecord(m->ev[(size_t)g][0], m->st[(size_t)g]));
    }
    for (int k = 0; k < iters; k++) {
        rc = multi_step(m, gather);
        if (rc) return rc;
        for (int g = 0; g < m->G; g++) { HIP_TRY(hipSetDevice(m->dev[(size_t)g])); HIP_TRY(hipEventRecord(m->ev[(size_t)g][(size_t)k + 1], m->st[(size_t)g])); }
    }
    rc = multi_sync(m);
    if (rc) return rc;
    std::vector<double> t((size_t)iters, 0.0);
    double              tot = 0;
    for (int g = 0; g < m->G; g++) {
        for (int k = 0; k < iters; k++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, m->ev[(size_t)g][(size_t)k], m->ev[(size_t)g][(size_t)k + 1]));
            t[(size_t)k] = std::max(t[(size_t)k], (double)ms * 1e-3);
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, m->ev[(size_t)g][0], m->ev[(size_t)g][(size_t)iters]));
        tot = std::max(tot, (double)ms * 1e-3);
    }
    double sum = 0;
    for (double v : t) sum += v;
    std::sort(t.begin(), t.end());
    *mean = sum / iters; *mn = t.front(); *mx = t.back(); *med = t[t.size() / 2]; *total = tot;
    return CVR_OK;
}

// device vectors, streams and communicators of a multi handle whose shards exist (m->H, m->dev, m->max_rows are set)
int multi_device_side(cvr_multi *m)
{
    const int G = m->G;
#define MD_TRY(expr) do { const int rc_ = (expr); if (rc_) return rc_; } while (0)
    // RCCL needs distinct devices; entries naming the same device are served by copies
    bool distinct = true;
    for (int a = 0; a < G; a++) for (int b = a + 1; b < G; b++) if (m->dev[(size_t)a] == m->dev[(size_t)b]) distinct = false;
    m->use_rccl = G > 1 && distinct;
    for (int g = 0; g < G; g++) {
        cvr_info info;
        MD_TRY(cvr_get_info(m->H[(size_t)g], &info));
        HIP_TRY(hipSetDevice(m->dev[(size_t)g]));
        const size_t ny = (size_t)std::max<int64_t>(info.yext_elems, m->max_rows), nx = (size_t)info.x_elems;
        HIP_TRY(hipMalloc(&m->dx[(size_t)g], m->vsz * std::max<size_t>(nx, 1)));
        HIP_TRY(hipMalloc(&m->dy[(size_t)g], m->vsz * std::max<size_t>(ny, 1)));
        HIP_TRY(hipMemset(m->dx[(size_t)g], 0, m->vsz * std::max<size_t>(nx, 1)));
        HIP_TRY(hipMemset(m->dy[(size_t)g], 0, m->vsz * std::max<size_t>(ny, 1)));
        if (G > 1) HIP_TRY(hipMalloc(&m->dall[(size_t)g], m->vsz * std::max<size_t>((size_t)G * (size_t)m->max_rows, 1)));
        HIP_TRY(hipStreamCreateWithFlags(&m->st[(size_t)g], hipStreamNonBlocking));
    }
    if (m->use_rccl) {
        const RcclApi *api = rccl_api();
        if (!api) { return fail(CVR_ERR_NO_DEVICE, "RCCL not found (librccl.so; set CVR_RCCL_LIB): %s", dlerror()); }
        const ncclResult_t r = api->comm_init_all(m->comm.data(), G, m->dev.data());
        if (r != ncclSuccess) { for (auto &c : m->comm) c = nullptr; return fail(CVR_ERR_HIP, "ncclCommInitAll: %s", api->error_string(r)); }
    }
#undef MD_TRY
    return CVR_OK;
}

}  // namespace

extern "C" {

int64_t cvr_row_partition(int64_t nrows, const int64_t *row_ptr, int32_t nparts, int64_t *bounds) { return cvr_row_partition_cost(nrows, row_ptr, nparts, 0, bounds); }

int64_t cvr_row_partition_cost(int64_t nrows, const int64_t *row_ptr, int32_t nparts, int32_t row_cost_milli, int64_t *bounds)
{
    if (nrows < 0 || nparts < 1 || row_cost_milli < 0 || !bounds || (nrows > 0 && !row_ptr)) return fail(CVR_ERR_INVALID, "bad partition arguments");
    const int64_t nz0 = nrows ? row_ptr[0] : 0;
    // cost of the rows before r, in thousandths of a non-zero: monotone in r, so the cut is a binary search (spmv.cpp:631-667 searches
    // row_ptr the same way, per thread, for non-zeros alone)
    auto cost = [&](int64_t r) { return (__int128)(row_ptr[r] - nz0) * 1000 + (__int128)r * row_cost_milli; };
    const int64_t nnz = nrows ? row_ptr[nrows] - nz0 : 0;
    bounds[0] = 0;
    bounds[nparts] = nrows;
    for (int32_t p = 1; p < nparts; p++) {
        // p / nparts of the non-zeros (rounded down, as the nnz-only rule always did) and of the rows' own cost
        const __int128 target = ((__int128)nnz * p / nparts) * 1000 + (__int128)nrows * row_cost_milli * p / nparts;
        int64_t        lo = 0, hi = nrows;            // the first row r in [0, nrows] with cost(r) >= target
        while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (cost(mid) >= target) hi = mid; else lo = mid + 1; }
        bounds[p] = std::min<int64_t>(std::max(lo, bounds[p - 1]), nrows);
    }
    int64_t most = 0;
    for (int32_t p = 0; p < nparts; p++) most = std::max(most, bounds[p + 1] - bounds[p]);
    return most;
}

int cvr_destroy_multi(cvr_multi *m)
{
    if (!m) return CVR_OK;
    const RcclApi *api = m->use_rccl ? rccl_api() : nullptr;
    for (int g = 0; g < m->G; g++) {
        (void)hipSetDevice(m->dev[(size_t)g]);
        if ((size_t)g < m->st.size() && m->st[(size_t)g]) (void)hipStreamSynchronize(m->st[(size_t)g]);
        if (api && (size_t)g < m->comm.size() && m->comm[(size_t)g]) (void)api->comm_destroy(m->comm[(size_t)g]);
        if ((size_t)g < m->ev.size()) for (hipEvent_t e : m->ev[(size_t)g]) (void)hipEventDestroy(e);
        for (std::vector<void *> *v : {&m->dx, &m->dy, &m->dall}) if ((size_t)g < v->size() && (*v)[(size_t)g]) (void)hipFree((*v)[(size_t)g]);
        if ((size_t)g < m->st.size() && m->st[(size_t)g]) (void)hipStreamDestroy(m->st[(size_t)g]);
        if ((size_t)g < m->H.size()) cvr_destroy(m->H[(size_t)g]);
    }
    delete m;
    return CVR_OK;
}

int cvr_create_multi(cvr_multi **out, const cvr_csr_view *csr, const cvr_options *opt_in, const int32_t *devices, int32_t ndevices)
{
    if (!out) return fail(CVR_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!csr || !devices || ndevices < 1 || ndevices > 64) return fail(CVR_ERR_INVALID, "bad multi-device arguments");
    if (csr->arrays_on_device) return fail(CVR_ERR_INVALID, "cvr_create_multi takes host arrays (one process, several devices)");
    int rc = check_csr(csr);
    if (rc) return rc;
    const int ndev = cvr_device_count();
    if (ndev <= 0) return fail(CVR_ERR_NO_DEVICE, "no HIP device visible: libcvr_amd has no CPU fallback");
    for (int g = 0; g < ndevices; g++)
        if (devices[g] < 0 || devices[g] >= ndev) return fail(CVR_ERR_NO_DEVICE, "device %d out of range [0, %d)", devices[g], ndev);
    cvr_multi *m = new (std::nothrow) cvr_multi;
    if (!m) return fail(CVR_ERR_NOMEM, "out of host memory");
    const int G = ndevices;
    m->G = G; m->f32 = csr->is_f32 != 0; m->vsz = m->f32 ? 4 : 8; m->nrows = csr->nrows; m->ncols = csr->ncols;
    m->dev.assign(devices, devices + G);
    m->bounds.assign((size_t)G + 1, 0);
    const char *pm = getenv("CVR_PARTITION");          // "nnz": balanced non-zeros, the reference's rule (spmv.cpp:584-627)
    m->max_rows = cvr_row_partition_cost(csr->nrows, csr->row_ptr, G, pm && !strcmp(pm, "nnz") ? 0 : CVR_ROW_COST_MILLI_DEFAULT, m->bounds.data());
    m->H.assign((size_t)G, nullptr); m->dx.assign((size_t)G, nullptr); m->dy.assign((size_t)G, nullptr); m->dall.assign((size_t)G, nullptr);
    m->st.assign((size_t)G, nullptr); m->comm.assign((size_t)G, nullptr); m->ev.resize((size_t)G);
#define MULTI_TRY(expr) do { const int rc_ = (expr); if (rc_) { cvr_destroy_multi(m); return rc_; } } while (0)
#define MULTI_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); cvr_destroy_multi(m); return CVR_ERR_HIP; } } while (0)
    std::vector<int64_t> lrp;
    for (int g = 0; g < G; g++) {
        // the shard as a matrix of its own: row pointers rebased to its first non-zero, columns and values from there on
        const int64_t b = m->bounds[(size_t)g], e = m->bounds[(size_t)g + 1], lo = csr->nrows ? csr->row_ptr[b] : 0;
        lrp.resize((size_t)(e - b) + 1);
        for (int64_t r = b; r <= e; r++) lrp[(size_t)(r - b)] = csr->nrows ? csr->row_ptr[r] - lo : 0;
        cvr_csr_view v = *csr;
        v.nrows = e - b; v.row_ptr = lrp.data();
        v.col_idx = csr->col_idx ? csr->col_idx + lo : nullptr;
        v.vals = csr->vals ? static_cast<const uint8_t *>(csr->vals) + (size_t)lo * m->vsz : nullptr;
        cvr_options o;
        if (opt_in) o = *opt_in; else cvr_default_options(&o);
        o.device = m->dev[(size_t)g];
        MULTI_TRY(cvr_create(&m->H[(size_t)g], &v, &o));
    }
    MULTI_TRY(multi_device_side(m));
#undef MULTI_TRY
#undef MULTI_HIP
    *out = m;
    return CVR_OK;
}

// shard handles that exist already (loaded from their image caches, or built by the caller): adopted, with everything else of
// cvr_create_multi.  bounds[n + 1], devices[n]; handle g must hold rows [bounds[g], bounds[g+1]) and be preprocessed.
int cvr_multi_from_handles(cvr_multi **out, cvr_handle **shards, const int64_t *bounds, const int32_t *devices, int32_t n)
{
    if (!out) return fail(CVR_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!shards || !bounds || !devices || n < 1 || n > 64) return fail(CVR_ERR_INVALID, "bad multi-device arguments");
    for (int g = 0; g < n; g++) {
        if (!shards[g] || !shards[g]->converted) return fail(CVR_ERR_STATE, "shard %d is missing or not preprocessed", g);
        if (shards[g]->info.nrows != bounds[g + 1] - bounds[g] || shards[g]->device != devices[g] || shards[g]->vsz != shards[0]->vsz || shards[g]->info.ncols != shards[0]->info.ncols)
            return fail(CVR_ERR_INVALID, "shard %d does not match its bounds / device / type", g);
    }
    cvr_multi *m = new (std::nothrow) cvr_multi;
    if (!m) return fail(CVR_ERR_NOMEM, "out of host memory");
    const int G = n;
    m->G = G; m->vsz = shards[0]->vsz; m->f32 = m->vsz == 4; m->nrows = bounds[n]; m->ncols = shards[0]->info.ncols;
    m->dev.assign(devices, devices + G);
    m->bounds.assign(bounds, bounds + G + 1);
    for (int g = 0; g < G; g++) m->max_rows = std::max(m->max_rows, bounds[g + 1] - bounds[g]);
    m->H.assign(shards, shards + G);
    m->dx.assign((size_t)G, nullptr); m->dy.assign((size_t)G, nullptr); m->dall.assign((size_t)G, nullptr);
    m->st.assign((size_t)G, nullptr); m->comm.assign((size_t)G, nullptr); m->ev.resize((size_t)G);
    const int rc = multi_device_side(m);
    if (rc) { for (auto &h : m->H) h = nullptr; cvr_destroy_multi(m); return rc; }      // (the caller keeps its handles on failure)
    m->converted = true;
    *out = m;
    return CVR_OK;
}

cvr_handle *cvr_multi_handle(cvr_multi *m, int32_t shard) { return m && shard >= 0 && shard < m->G ? m->H[(size_t)shard] : nullptr; }

int cvr_preprocess_multi(cvr_multi *m, int keep_csr, double *seconds)
{
    if (!m) return fail(CVR_ERR_INVALID, "handle is null");
    double worst = 0;
    for (int g = 0; g < m->G; g++) {
        double   s = 0;
        const int rc = cvr_preprocess(m->H[(size_t)g], keep_csr, &s);      // (spmv.cpp:1857, per shard)
        if (rc) return rc;
        worst = std::max(worst, s + m->H[(size_t)g]->info.plan_s);
    }
    if (seconds) *seconds = worst;
    m->converted = true;
    return CVR_OK;
}

int cvr_spmv_multi(cvr_multi *m, const void *x_host, void *y_host, int iters, cvr_timing *tm)
{
    if (!m || !x_host || !y_host) return fail(CVR_ERR_INVALID, "null argument");
    if (!m->converted) return fail(CVR_ERR_STATE, "cvr_spmv_multi before cvr_preprocess_multi");
    if (iters < 1) iters = 1;
    double t0 = now_s();
    for (int g = 0; g < m->G; g++) {                       // x replicated
        HIP_TRY(hipSetDevice(m->dev[(size_t)g]));
        if (m->ncols) HIP_TRY(hipMemcpyAsync(m->dx[(size_t)g], x_host, m->vsz * (size_t)m->ncols, hipMemcpyHostToDevice, m->st[(size_t)g]));
    }
    int rc = multi_sync(m);
    if (rc) return rc;
    const double h2d = now_s() - t0;
    cvr_timing t;
    memset(&t, 0, sizeof(t));
    t.iters = iters; t.h2d_s = h2d;
    rc = multi_timed(m, false, iters, &t.mean_s, &t.min_s, &t.median_s, &t.max_s, &t.total_s);      // the SpMVs alone
    if (rc) return rc;
    if (m->G > 1) {
        double tot = 0;
        rc = multi_timed(m, true, iters, &t.step_mean_s, &t.step_min_s, &t.step_median_s, &t.step_max_s, &tot);
        if (rc) return rc;
        t.gather_mean_s = std::max(0.0, t.step_mean_s - t.mean_s);
    } else {
        t.step_mean_s = t.mean_s; t.step_min_s = t.min_s; t.step_median_s = t.median_s; t.step_max_s = t.max_s;
    }
    t0 = now_s();
    for (int g = 0; g < m->G; g++) {                       // y: from the first device's gathered copy (one shard: its y)
        const int64_t b = m->bounds[(size_t)g], n = m->bounds[(size_t)g + 1] - b;
        if (!n) continue;
        HIP_TRY(hipSetDevice(m->dev[0]));
        const uint8_t *src = m->G > 1 ? static_cast<const uint8_t *>(m->dall[0]) + (size_t)g * (size_t)m->max_rows * m->vsz : static_cast<const uint8_t *>(m->dy[0]);
        HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(y_host) + (size_t)b * m->vsz, src, m->vsz * (size_t)n, hipMemcpyDeviceToHost, m->st[0]));
    }
    HIP_TRY(hipSetDevice(m->dev[0]));
    HIP_TRY(hipStreamSynchronize(m->st[0]));
    t.d2h_s = now_s() - t0;
    if (tm) *tm = t;
    return CVR_OK;
}

int cvr_multi_shards(const cvr_multi *m) { return m ? m->G : 0; }
int cvr_multi_uses_rccl(const cvr_multi *m) { return m && m->use_rccl ? 1 : 0; }

int cvr_multi_info(const cvr_multi *m, int32_t shard, cvr_info *info, int64_t *row_begin, int64_t *row_end, int32_t *device)
{
    if (!m || shard < 0 || shard >= m->G) return fail(CVR_ERR_INVALID, "shard out of range");
    if (info) { const int rc = cvr_get_info(m->H[(size_t)shard], info); if (rc) return rc; }
    if (row_begin) *row_begin = m->bounds[(size_t)shard];
    if (row_end) *row_end = m->bounds[(size_t)shard + 1];
    if (device) *device = m->dev[(size_t)shard];
    return CVR_OK;
}

}  // extern "C"

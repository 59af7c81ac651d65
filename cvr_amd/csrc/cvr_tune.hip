// cvr_tune.hip -- the layout by measurement: cvr_tune / cvr_tune_steps build the candidates for real and time them.
#include "cvr_internal.h"

using namespace cvrh;

extern "C" {

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
        const int     P = (int)std::min(32.0, std::max(2.0, std::floor(xbytes / 450e3 + 0.5)));
        for (int w : {8, 7, 4}) {
            int S = (int)std::ceil(slots / (64.0 * w * (double)(chip_of(opt.device).cus - 4)) / 4.0) * 4;
            if (S < 8) S = 8;
            if (S > 128) continue;
            for (int win : {(int)(98304 / vs), 0})
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

}  // extern "C"

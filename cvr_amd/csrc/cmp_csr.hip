// cmp_csr.hip -- GPU-resident CSR comparators for the preprocessing-amortisation report (SURVEY.md 8(f) item 2;
// paper Eq. 1: I_pre = T_pre / (T_baseline - T_new), with MKL's CSR as the baseline on KNL).  NOT part of the
// product path and not an oracle: a plain CSR-vector kernel (the reference's CSR loop, spmv.cpp:1843-1850, with
// L lanes per row) and rocSPARSE's CSR SpMV, timed beside the CVR64 kernel on the same device arrays.
// Built into its own library (libcvr_cmp.so) so that libcvr_amd.so does not depend on rocSPARSE.
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

thread_local char g_err[256] = "";
#define HIPC(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", #e, hipGetErrorString(e_)); return -3; } } while (0)
#define RSC(e)  do { rocsparse_status e_ = (e); if (e_ != rocsparse_status_success) { snprintf(g_err, sizeof(g_err), "%s: rocsparse status %d", #e, (int)e_); return -3; } } while (0)

template <typename T, int L>
__global__ __launch_bounds__(256) void csr_vector_kernel(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                          const T *__restrict__ va, const T *__restrict__ x, T *__restrict__ y, int nrows)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = tid / L, sub = tid % L;
    T         sum = 0;
    if (row < nrows) {
        const int b = rp[row], e = rp[row + 1];
        for (int j = b + sub; j < e; j += L) sum = fma(va[j], x[ci[j]], sum);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, L);
    if (row < nrows && sub == 0) y[row] = sum;
}

template <typename T>
void launch_vector(int L, const int32_t *rp, const int32_t *ci, const T *va, const T *x, T *y, int nrows, hipStream_t st)
{
    const long long threads = (long long)nrows * L;
    const dim3      grid((unsigned)((threads + 255) / 256)), block(256);
    switch (L) {
    case 1:  hipLaunchKernelGGL((csr_vector_kernel<T, 1>), grid, block, 0, st, rp, ci, va, x, y, nrows); break;
    case 2:  hipLaunchKernelGGL((csr_vector_kernel<T, 2>), grid, block, 0, st, rp, ci, va, x, y, nrows); break;
    case 4:  hipLaunchKernelGGL((csr_vector_kernel<T, 4>), grid, block, 0, st, rp, ci, va, x, y, nrows); break;
    case 8:  hipLaunchKernelGGL((csr_vector_kernel<T, 8>), grid, block, 0, st, rp, ci, va, x, y, nrows); break;
    case 16: hipLaunchKernelGGL((csr_vector_kernel<T, 16>), grid, block, 0, st, rp, ci, va, x, y, nrows); break;
    case 32: hipLaunchKernelGGL((csr_vector_kernel<T, 32>), grid, block, 0, st, rp, ci, va, x, y, nrows); break;
    default: hipLaunchKernelGGL((csr_vector_kernel<T, 64>), grid, block, 0, st, rp, ci, va, x, y, nrows); break;
    }
}

}  // namespace

struct cmp_csr {
    int          nrows = 0, ncols = 0, f32 = 0;
    long long    nnz = 0;
    int32_t     *rp = nullptr, *ci = nullptr;
    void        *va = nullptr, *x = nullptr, *y = nullptr;
    hipStream_t  st = nullptr;
    hipEvent_t   e0 = nullptr, e1 = nullptr;
    rocsparse_handle      rs = nullptr;
    rocsparse_spmat_descr mat = nullptr;
    rocsparse_dnvec_descr vx = nullptr, vy = nullptr;
    void        *buf = nullptr;
    size_t       bufsz = 0;
    int          prepared_alg = -1;
    double       rs_preprocess_s = 0;
};

extern "C" {

const char *cmp_last_error(void) { return g_err; }

int cmp_csr_create(cmp_csr **out, long long nrows, long long ncols, const int64_t *rp, const int32_t *ci, const void *va, int f32, int device)
{
    *out = nullptr;
    const long long nnz = rp[nrows];
    if (nnz >= 0x7fffffffLL || nrows >= 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "comparator handles nnz < 2^31 only"); return -1; }
    cmp_csr *c = new cmp_csr;
    c->nrows = (int)nrows; c->ncols = (int)ncols; c->nnz = nnz; c->f32 = f32;
    const size_t vs = f32 ? 4 : 8;
    std::vector<int32_t> rp32((size_t)nrows + 1);
    for (long long i = 0; i <= nrows; i++) rp32[(size_t)i] = (int32_t)rp[i];
    HIPC(hipSetDevice(device));
    HIPC(hipStreamCreate(&c->st));
    HIPC(hipEventCreate(&c->e0));
    HIPC(hipEventCreate(&c->e1));
    HIPC(hipMalloc(&c->rp, 4 * ((size_t)nrows + 1)));
    HIPC(hipMalloc(&c->ci, 4 * (size_t)(nnz ? nnz : 1)));
    HIPC(hipMalloc(&c->va, vs * (size_t)(nnz ? nnz : 1)));
    HIPC(hipMalloc(&c->x, vs * (size_t)(ncols + 1)));
    HIPC(hipMalloc(&c->y, vs * (size_t)(nrows + 1)));
    HIPC(hipMemcpy(c->rp, rp32.data(), 4 * ((size_t)nrows + 1), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(c->ci, ci, 4 * (size_t)nnz, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(c->va, va, vs * (size_t)nnz, hipMemcpyHostToDevice));
    HIPC(hipMemset(c->x, 0, vs * (size_t)(ncols + 1)));
    HIPC(hipMemset(c->y, 0, vs * (size_t)(nrows + 1)));
    RSC(rocsparse_create_handle(&c->rs));
    RSC(rocsparse_set_stream(c->rs, c->st));
    const rocsparse_datatype dt = f32 ? rocsparse_datatype_f32_r : rocsparse_datatype_f64_r;
    RSC(rocsparse_create_csr_descr(&c->mat, nrows, ncols, nnz, c->rp, c->ci, c->va, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                   rocsparse_index_base_zero, dt));
    RSC(rocsparse_create_dnvec_descr(&c->vx, ncols, c->x, dt));
    RSC(rocsparse_create_dnvec_descr(&c->vy, nrows, c->y, dt));
    *out = c;
    return 0;
}

int cmp_csr_set_x(cmp_csr *c, const void *x_host)
{
    HIPC(hipMemcpy(c->x, x_host, (c->f32 ? 4 : 8) * (size_t)c->ncols, hipMemcpyHostToDevice));
    return 0;
}

int cmp_csr_get_y(cmp_csr *c, void *y_host)
{
    HIPC(hipStreamSynchronize(c->st));
    HIPC(hipMemcpy(y_host, c->y, (c->f32 ? 4 : 8) * (size_t)c->nrows, hipMemcpyDeviceToHost));
    return 0;
}

// kind: 0 = own CSR-vector kernel (L lanes per row from the mean row length), 1 = rocSPARSE default (adaptive),
//       2 = rocSPARSE rowsplit, 3 = rocSPARSE LRB.  One launch on the comparator's stream.
static int run_once(cmp_csr *c, int kind)
{
    if (kind == 0) {
        int       L = 1;
        const double mean = c->nrows ? (double)c->nnz / c->nrows : 0;
        while (L < 64 && L < mean) L *= 2;
        if (c->f32) launch_vector<float>(L, c->rp, c->ci, (const float *)c->va, (const float *)c->x, (float *)c->y, c->nrows, c->st);
        else launch_vector<double>(L, c->rp, c->ci, (const double *)c->va, (const double *)c->x, (double *)c->y, c->nrows, c->st);
        HIPC(hipGetLastError());
        return 0;
    }
    const rocsparse_spmv_alg alg = kind == 2 ? rocsparse_spmv_alg_csr_rowsplit : kind == 3 ? rocsparse_spmv_alg_csr_lrb : rocsparse_spmv_alg_csr_adaptive;
    const rocsparse_datatype dt = c->f32 ? rocsparse_datatype_f32_r : rocsparse_datatype_f64_r;
    const double a64 = 1, b64 = 0;
    const float  a32 = 1, b32 = 0;
    const void  *al = c->f32 ? (const void *)&a32 : (const void *)&a64, *be = c->f32 ? (const void *)&b32 : (const void *)&b64;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wdeprecated-declarations"
    if (c->prepared_alg != kind) {
        size_t sz = 0;
        RSC(rocsparse_spmv(c->rs, rocsparse_operation_none, al, c->mat, c->vx, be, c->vy, dt, alg, rocsparse_spmv_stage_buffer_size, &sz, nullptr));
        if (sz > c->bufsz) { if (c->buf) (void)hipFree(c->buf); HIPC(hipMalloc(&c->buf, sz ? sz : 16)); c->bufsz = sz; }
        HIPC(hipStreamSynchronize(c->st));
        HIPC(hipEventRecord(c->e0, c->st));
        RSC(rocsparse_spmv(c->rs, rocsparse_operation_none, al, c->mat, c->vx, be, c->vy, dt, alg, rocsparse_spmv_stage_preprocess, &sz, c->buf));
        HIPC(hipEventRecord(c->e1, c->st));
        HIPC(hipEventSynchronize(c->e1));
        float ms = 0;
        HIPC(hipEventElapsedTime(&ms, c->e0, c->e1));
        c->rs_preprocess_s = ms * 1e-3;
        c->prepared_alg = kind;
    }
    size_t sz = c->bufsz;
    RSC(rocsparse_spmv(c->rs, rocsparse_operation_none, al, c->mat, c->vx, be, c->vy, dt, alg, rocsparse_spmv_stage_compute, &sz, c->buf));
#pragma clang diagnostic pop
    return 0;
}

int cmp_csr_bench(cmp_csr *c, int kind, int warmup, int iters, double *mean_s, double *preprocess_s)
{
    for (int i = 0; i < warmup; i++) { int rc = run_once(c, kind); if (rc) return rc; }
    HIPC(hipEventRecord(c->e0, c->st));
    for (int i = 0; i < iters; i++) { int rc = run_once(c, kind); if (rc) return rc; }
    HIPC(hipEventRecord(c->e1, c->st));
    HIPC(hipEventSynchronize(c->e1));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, c->e0, c->e1));
    if (mean_s) *mean_s = ms * 1e-3 / iters;
    if (preprocess_s) *preprocess_s = kind == 0 ? 0.0 : c->rs_preprocess_s;
    return 0;
}

int cmp_csr_destroy(cmp_csr *c)
{
    if (!c) return 0;
    if (c->vx) rocsparse_destroy_dnvec_descr(c->vx);
    if (c->vy) rocsparse_destroy_dnvec_descr(c->vy);
    if (c->mat) rocsparse_destroy_spmat_descr(c->mat);
    if (c->rs) rocsparse_destroy_handle(c->rs);
    for (void *p : {(void *)c->rp, (void *)c->ci, c->va, c->x, c->y, c->buf}) if (p) (void)hipFree(p);
    if (c->e0) (void)hipEventDestroy(c->e0);
    if (c->e1) (void)hipEventDestroy(c->e1);
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
    return 0;
}

}  // extern "C"

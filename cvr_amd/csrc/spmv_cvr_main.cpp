// spmv_cvr_main.cpp -- the host program: same command line and same greppable output as the reference
//
//     ./spmv.cvr [matrix.mtx] [nThreads] [nIters]          (main, /root/reference/spmv.cpp:1675-1948;
//                                                           argv: :1693, :1703, :1771; README.md:26-28, 47-49)
//
// but the hot path (pre_processing, spmv.cpp:1857; spmv_compute_kernel, spmv.cpp:1882) runs on MI355X GPUs
// through the C ABI of include/cvr_amd.h.  nThreads keeps its meaning for the host side (the CSR self-check
// loop).  Everything else is selected by environment variables so that the CLI stays drop-in:
//     CVR_DEVICES=0,1,..  GPUs to shard the rows over (default 0); x replicated, y all-gathered by RCCL
//                         (the same GPU listed several times: same sharding, slices moved by D2D copies)
//     CVR_X=ones|rand     x = 1.0 as the reference (spmv.cpp:556-563) or the seeded non-constant x
//     CVR_MM=refcompat|strict   loader mode (default refcompat = the reference loader's arrays)
//     CVR_S=<steps>       lane-stream length per chunk (default: chosen from the matrix size)
//     CVR_CACHE=1         keep / reuse a binary image of the parsed matrix (<mtx>.ref.cvrbin / .strict.cvrbin)
// Exit code 0 as the reference (spmv.cpp:1947), 1 on loader errors (spmv.cpp:325-355), 2 on usage / device errors.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cvr_amd.h"

#define HIP_OK(e)  do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define NCCL_OK(e) do { ncclResult_t e_ = (e); if (e_ != ncclSuccess) { fprintf(stderr, "RCCL error %s at %s:%d\n", ncclGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CVR_OKAY(e) do { int e_ = (e); if (e_ != CVR_OK) { fprintf(stderr, "cvr error %d: %s (%s:%d)\n", e_, cvr_last_error(), __FILE__, __LINE__); return 2; } } while (0)

static std::vector<int> parse_devices(const char *s)
{
    std::vector<int> d;
    if (!s || !*s) { d.push_back(0); return d; }
    std::string t(s);
    size_t      p = 0;
    while (p < t.size()) {
        size_t q = t.find(',', p);
        if (q == std::string::npos) q = t.size();
        d.push_back(atoi(t.substr(p, q - p).c_str()));
        p = q + 1;
    }
    return d;
}

int main(int argc, char **argv)
{
    if (argc < 4) {   // the reference segfaults here (no argc check); a usage line is a superset
        fprintf(stderr, "usage: %s [matrix.mtx] [nThreads] [nIters]\n", argv[0]);
        return 2;
    }
    const char *fn = argv[1];
    int         nthreads = atoi(argv[2]);   // spmv.cpp:1693
    int         niters = atoi(argv[3]);     // spmv.cpp:1703
    if (nthreads < 1) nthreads = 1;
    if (niters < 1) niters = 1;
    const char       *xenv = getenv("CVR_X"), *mmenv = getenv("CVR_MM"), *senv = getenv("CVR_S");
    const int         xmode = xenv && !strcmp(xenv, "rand") ? 1 : 0;
    const int         mmmode = mmenv && !strcmp(mmenv, "strict") ? CVR_MM_STRICT : CVR_MM_REFCOMPAT;
    std::vector<int>  devs = parse_devices(getenv("CVR_DEVICES"));
    const int         G = (int)devs.size();

    // CVR_CACHE=1: keep / reuse a binary image of the parsed matrix next to the .mtx file
    cvr_mm_matrix     m;
    const std::string bin = std::string(fn) + (mmmode == CVR_MM_STRICT ? ".strict.cvrbin" : ".ref.cvrbin");
    int               rc = CVR_ERR_IO;
    const bool        use_cache = getenv("CVR_CACHE") && atoi(getenv("CVR_CACHE"));
    if (use_cache) rc = cvr_mm_read_bin(bin.c_str(), &m);
    if (rc) {
        rc = cvr_mm_read(fn, mmmode, &m);   // spmv.cpp:1771
        if (!rc && use_cache) (void)cvr_mm_write_bin(bin.c_str(), &m);
    }
    if (rc) { fprintf(stderr, "Error: unable to read matrix file %s (%d)\n", fn, rc); return 1; }
    printf("Matrix %s: %lld rows, %lld columns, %lld stored entries (%s loader)\n", fn, (long long)m.ref_numRows,
           (long long)m.ref_numCols, (long long)m.ref_nItems, mmmode == CVR_MM_STRICT ? "strict" : "reference-compatible");

    std::vector<double> x((size_t)m.ncols + 1), yref((size_t)m.nrows + 1, 0.0), y((size_t)m.nrows + 1, 0.0);
    cvr_fill_x(x.data(), m.ncols, xmode);   // spmv.cpp:1788
    cvr_csr_spmv_host(m.nrows, m.row_ptr, m.col_idx, m.vals, x.data(), yref.data(), nthreads);   // spmv.cpp:1843-1850

    // rows sharded over the GPUs: contiguous blocks with balanced nnz, cut at row boundaries
    // (the reference balances nnz per thread the same way, spmv.cpp:584-667)
    std::vector<int64_t> bounds((size_t)G + 1, 0);
    bounds[(size_t)G] = m.nrows;
    for (int g = 1; g < G; g++) {
        const int64_t target = m.row_ptr[0] + (m.row_ptr[m.nrows] - m.row_ptr[0]) / G * g;
        bounds[(size_t)g] = std::lower_bound(m.row_ptr, m.row_ptr + m.nrows + 1, target) - m.row_ptr;
        bounds[(size_t)g] = std::min<int64_t>(std::max(bounds[(size_t)g], bounds[(size_t)g - 1]), m.nrows);
    }
    int64_t max_rows = 0;
    for (int g = 0; g < G; g++) max_rows = std::max(max_rows, bounds[(size_t)g + 1] - bounds[(size_t)g]);

    std::vector<cvr_handle *> H((size_t)G, nullptr);
    std::vector<cvr_info>     info((size_t)G);
    std::vector<std::vector<int64_t>> lrp((size_t)G);
    double pre_s = 0;
    for (int g = 0; g < G; g++) {
        const int64_t b = bounds[(size_t)g], e = bounds[(size_t)g + 1], lo = m.row_ptr[b];
        lrp[(size_t)g].resize((size_t)(e - b) + 1);
        for (int64_t r = b; r <= e; r++) lrp[(size_t)g][(size_t)(r - b)] = m.row_ptr[r] - lo;
        cvr_csr_view v = {};
        v.nrows = e - b; v.ncols = m.ncols; v.row_ptr = lrp[(size_t)g].data(); v.col_idx = m.col_idx + lo; v.vals = m.vals + lo; v.is_f32 = 0;
        cvr_options o;
        cvr_default_options(&o);
        o.device = devs[(size_t)g];
        if (senv) o.steps_per_chunk = atoi(senv);
        CVR_OKAY(cvr_create(&H[(size_t)g], &v, &o));
        double s = 0;
        CVR_OKAY(cvr_preprocess(H[(size_t)g], 0, &s));   // spmv.cpp:1857
        CVR_OKAY(cvr_get_info(H[(size_t)g], &info[(size_t)g]));
        pre_s = std::max(pre_s, s + info[(size_t)g].plan_s);
    }
    printf("The Pre-processing(CSR->CVR)   Time of CVR   is %g seconds.   [file: %s] [threads: %d]\n", pre_s, fn, nthreads);   // spmv.cpp:1009

    // device vectors: x replicated; per GPU its y_ext; with G > 1 a gathered y of G * max_rows
    std::vector<double *>    dx((size_t)G), dy((size_t)G), dall((size_t)G, nullptr);
    std::vector<hipStream_t> st((size_t)G);
    std::vector<hipEvent_t>  e0((size_t)G), e1((size_t)G);
    std::vector<ncclComm_t>  comm((size_t)G);
    for (int g = 0; g < G; g++) {
        HIP_OK(hipSetDevice(devs[(size_t)g]));
        const size_t ny = (size_t)std::max<int64_t>(info[(size_t)g].yext_elems, max_rows);
        HIP_OK(hipMalloc(&dx[(size_t)g], sizeof(double) * (size_t)info[(size_t)g].x_elems));
        HIP_OK(hipMalloc(&dy[(size_t)g], sizeof(double) * ny));
        HIP_OK(hipMemset(dx[(size_t)g], 0, sizeof(double) * (size_t)info[(size_t)g].x_elems));
        HIP_OK(hipMemset(dy[(size_t)g], 0, sizeof(double) * ny));
        HIP_OK(hipMemcpy(dx[(size_t)g], x.data(), sizeof(double) * (size_t)m.ncols, hipMemcpyHostToDevice));
        if (G > 1) HIP_OK(hipMalloc(&dall[(size_t)g], sizeof(double) * (size_t)G * (size_t)max_rows));
        HIP_OK(hipStreamCreateWithFlags(&st[(size_t)g], hipStreamNonBlocking));
        HIP_OK(hipEventCreate(&e0[(size_t)g]));
        HIP_OK(hipEventCreate(&e1[(size_t)g]));
    }
    // RCCL needs distinct devices.  CVR_DEVICES=0,0,.. (the same GPU listed several times) keeps the sharding, the
    // per-shard handles and the gather layout but moves the slices with device-to-device copies: a plumbing check
    // of the multi-GPU path on a 1-GPU box.
    bool distinct = true;
    for (int a = 0; a < G; a++) for (int b2 = a + 1; b2 < G; b2++) if (devs[(size_t)a] == devs[(size_t)b2]) distinct = false;
    const bool use_rccl = G > 1 && distinct;
    if (use_rccl) NCCL_OK(ncclCommInitAll(comm.data(), G, devs.data()));

    auto spmv_all = [&](bool gather) -> int {
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[(size_t)g]));
            CVR_OKAY(cvr_spmv_device(H[(size_t)g], dx[(size_t)g], dy[(size_t)g], st[(size_t)g]));   // spmv.cpp:1882
        }
        if (gather && G > 1 && !use_rccl) {
            for (int g = 0; g < G; g++) {          // every "rank" receives every slice, as the all-gather would deliver
                HIP_OK(hipSetDevice(devs[(size_t)g]));
                for (int src = 0; src < G; src++)
                    HIP_OK(hipMemcpyAsync(dall[(size_t)g] + (size_t)src * (size_t)max_rows, dy[(size_t)src], sizeof(double) * (size_t)max_rows,
                                          hipMemcpyDeviceToDevice, st[(size_t)src]));
            }
        }
        if (gather && use_rccl) {
            NCCL_OK(ncclGroupStart());
            for (int g = 0; g < G; g++)
                NCCL_OK(ncclAllGather(dy[(size_t)g], dall[(size_t)g], (size_t)max_rows, ncclDouble, comm[(size_t)g], st[(size_t)g]));
            NCCL_OK(ncclGroupEnd());
        }
        return 0;
    };
    auto sync_all = [&]() -> int {
        for (int g = 0; g < G; g++) { HIP_OK(hipSetDevice(devs[(size_t)g])); HIP_OK(hipStreamSynchronize(st[(size_t)g])); }
        return 0;
    };
    auto timed = [&](bool gather, double *sec) -> int {
        for (int w = 0; w < 10; w++) if (spmv_all(gather)) return 2;
        if (sync_all()) return 2;
        for (int g = 0; g < G; g++) { HIP_OK(hipSetDevice(devs[(size_t)g])); HIP_OK(hipEventRecord(e0[(size_t)g], st[(size_t)g])); }
        for (int k = 0; k < niters; k++) if (spmv_all(gather)) return 2;
        for (int g = 0; g < G; g++) { HIP_OK(hipSetDevice(devs[(size_t)g])); HIP_OK(hipEventRecord(e1[(size_t)g], st[(size_t)g])); }
        if (sync_all()) return 2;
        double worst = 0;
        for (int g = 0; g < G; g++) { float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0[(size_t)g], e1[(size_t)g])); worst = std::max(worst, (double)ms * 1e-3 / niters); }
        *sec = worst;
        return 0;
    };
    double t_compute = 0, t_total = 0;
    if (timed(false, &t_compute)) return 2;
    t_total = t_compute;
    if (G > 1 && timed(true, &t_total)) return 2;

    // y back for the verdict (from the gathered copy of GPU 0 when sharded)
    for (int g = 0; g < G; g++) {
        const int64_t b = bounds[(size_t)g], n = bounds[(size_t)g + 1] - b;
        HIP_OK(hipSetDevice(devs[G > 1 ? 0 : (size_t)g]));
        const double *src = G > 1 ? dall[0] + (size_t)g * (size_t)max_rows : dy[(size_t)g];
        if (n) HIP_OK(hipMemcpy(y.data() + b, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    }

    const double nItems = (double)m.ref_nItems;
    printf("The SpMV Execution Time of CVR    is %g seconds.   [file: %s] [threads: %d]\n", t_total, fn, nthreads);   // spmv.cpp:1662
    printf("         The Throughput of CVR    is %g GFlops.    [file: %s] [threads: %d]\n", nItems / t_total / 1e9, fn, nthreads);   // spmv.cpp:1664
    // the reference checks rows 0..numRows-1 of its 1-based arrays (spmv.cpp:1920): the same rows here
    const int64_t checked = mmmode == CVR_MM_REFCOMPAT ? m.ref_numRows : m.nrows;
    const int64_t wrong = cvr_verdict(y.data(), yref.data(), checked);
    if (!wrong) printf("     Very Good! Your result is correct  \n");                       // spmv.cpp:1932
    else printf("Warning: %lld out of %lld is wrong\n", (long long)wrong, (long long)m.ref_nItems);   // spmv.cpp:1935

    const double nnz_true = (double)m.ref_nItemsRaw;
    const double balg = nnz_true * 12.0 + ((double)m.ref_numRows + 1) * 4.0 + (double)m.ref_numCols * 8.0 + (double)m.ref_numRows * 8.0;
    int64_t      chunks = 0, cut = 0;
    for (int g = 0; g < G; g++) { chunks += info[(size_t)g].nchunks; cut += info[(size_t)g].nshared; }
    printf("{\"backend\":\"hip-gfx950\",\"gpus\":%d,\"iters\":%d,\"nnz\":%.0f,\"rows\":%lld,\"steps_per_chunk\":%d,\"chunks\":%lld,\"rows_cut\":%lld,"
           "\"preprocess_s\":%.6g,\"spmv_compute_s\":%.6g,\"spmv_with_gather_s\":%.6g,\"gflops_2nnz\":%.6g,\"gbs_alg\":%.6g,"
           "\"frac_of_8TBs_per_gpu\":%.4f,\"wrong\":%lld}\n",
           G, niters, nnz_true, (long long)m.ref_numRows, info[0].steps_per_chunk, (long long)chunks, (long long)cut, pre_s, t_compute, t_total,
           2.0 * nnz_true / t_total / 1e9, balg / t_total / 1e9, balg / t_compute / (8e12 * G), (long long)wrong);

    // CVR_POWER=<iterations>: the iterative caller on the same handle (one GPU, square matrix): x <- A x / ||A x||
    const char *penv = getenv("CVR_POWER");
    if (penv && atoi(penv) > 0 && G == 1 && m.nrows == m.ncols) {
        std::vector<double> ones((size_t)m.ncols, 1.0);
        HIP_OK(hipMemcpy(dx[0], ones.data(), sizeof(double) * (size_t)m.ncols, hipMemcpyHostToDevice));
        double lambda = 0, sec = 0;
        CVR_OKAY(cvr_power_iteration(H[0], nullptr, nullptr, atoi(penv), dx[0], &lambda, &sec, st[0]));
        printf("{\"power_iterations\":%d,\"rayleigh_quotient\":%.15g,\"seconds_per_iteration\":%.6g}\n", atoi(penv), lambda, sec);
    }

    for (int g = 0; g < G; g++) {
        (void)hipSetDevice(devs[(size_t)g]);
        if (use_rccl) ncclCommDestroy(comm[(size_t)g]);
        (void)hipFree(dx[(size_t)g]); (void)hipFree(dy[(size_t)g]);
        if (dall[(size_t)g]) (void)hipFree(dall[(size_t)g]);
        cvr_destroy(H[(size_t)g]);
    }
    cvr_mm_free(&m);
    return 0;
}

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
//     CVR_CACHE=1         caches keyed to the .mtx file beside it: the parsed CSR (.csrbin) and, on one GPU, the converted image (.cvrimg)
// Exit code 0 as the reference (spmv.cpp:1947), 1 on loader errors (spmv.cpp:325-355), 2 on usage / device errors.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cvr_amd.h"

#define HIP_OK(e)  do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
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

    // CVR_CACHE=1: caches beside the .mtx file, both keyed to it (size, mtime, hash of the first and last MiB: a changed file is
    // parsed and converted again): the parsed CSR (<mtx>.ref.csrbin / .strict.csrbin) and, on one GPU, the converted CVR64 image
    // (<mtx>.ref.cvrimg / .strict.cvrimg, keyed by the options and the device geometry as well)
    cvr_mm_matrix     m;
    const bool        use_cache = getenv("CVR_CACHE") && atoi(getenv("CVR_CACHE"));
    int               csr_hit = 0;
    int               rc = use_cache ? cvr_mm_read_cached(fn, mmmode, &m, &csr_hit) : cvr_mm_read(fn, mmmode, &m);   // spmv.cpp:1771
    if (rc) { fprintf(stderr, "Error: unable to read matrix file %s (%d)\n", fn, rc); return 1; }
    printf("Matrix %s: %lld rows, %lld columns, %lld stored entries (%s loader)\n", fn, (long long)m.ref_numRows,
           (long long)m.ref_numCols, (long long)m.ref_nItems, mmmode == CVR_MM_STRICT ? "strict" : "reference-compatible");

    std::vector<double> x((size_t)m.ncols + 1), yref((size_t)m.nrows + 1, 0.0), y((size_t)m.nrows + 1, 0.0);
    cvr_fill_x(x.data(), m.ncols, xmode);   // spmv.cpp:1788
    cvr_csr_spmv_host(m.nrows, m.row_ptr, m.col_idx, m.vals, x.data(), yref.data(), nthreads);   // spmv.cpp:1843-1850

    // One call = all GPUs (include/cvr_amd.h, cvr_create_multi): the library cuts the rows into one block per device (balanced
    // non-zeros, at row boundaries), builds the shards, replicates x and all-gathers y -- this program only names the devices,
    // as the reference's main only names the thread count (spmv.cpp:1857, 1882).
    cvr_csr_view view = {};
    view.nrows = m.nrows; view.ncols = m.ncols; view.row_ptr = m.row_ptr; view.col_idx = m.col_idx; view.vals = m.vals; view.is_f32 = 0;
    cvr_options opt;
    cvr_default_options(&opt);
    if (senv) opt.steps_per_chunk = atoi(senv);
    std::vector<int32_t> dev32(devs.begin(), devs.end());
    cvr_multi *M = nullptr;
    double     pre_s = 0;
    const char *image_cache = "off";
    cvr_source_key key;
    const std::string img = std::string(fn) + (mmmode == CVR_MM_STRICT ? ".strict.cvrimg" : ".ref.cvrimg");
    const bool        try_image = use_cache && G == 1 && cvr_source_key_of(fn, mmmode, &key) == CVR_OK;
    if (try_image) {      // the converted image from an earlier run: no analysis, no planner, no converter
        cvr_handle *H = nullptr;
        opt.device = devs[0];
        double load_s = 0;
        if (cvr_load_image(&H, img.c_str(), &key, &opt, &load_s) == CVR_OK) {
            const int64_t b2[2] = {0, m.nrows};
            CVR_OKAY(cvr_multi_from_handles(&M, &H, b2, dev32.data(), 1));
            pre_s = load_s;
            image_cache = "hit";
        } else image_cache = "miss";
    }
    if (!M) {
        CVR_OKAY(cvr_create_multi(&M, &view, &opt, dev32.data(), G));
        CVR_OKAY(cvr_preprocess_multi(M, 0, &pre_s));   // spmv.cpp:1857
        if (try_image && cvr_save_image(cvr_multi_handle(M, 0), img.c_str(), &key) != CVR_OK) fprintf(stderr, "note: image cache not written: %s\n", cvr_last_error());
    }
    printf("The Pre-processing(CSR->CVR)   Time of CVR   is %g seconds.   [file: %s] [threads: %d]\n", pre_s, fn, nthreads);   // spmv.cpp:1009

    cvr_timing tm;
    CVR_OKAY(cvr_spmv_multi(M, x.data(), y.data(), niters, &tm));   // spmv.cpp:1882
    const double t_compute = tm.mean_s, t_total = tm.step_mean_s;

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
    int          S0 = 0;
    for (int g = 0; g < G; g++) {
        cvr_info info;
        CVR_OKAY(cvr_multi_info(M, g, &info, nullptr, nullptr, nullptr));
        chunks += info.nchunks; cut += info.nshared;
        if (g == 0) S0 = info.steps_per_chunk;
    }
    printf("{\"backend\":\"hip-gfx950\",\"gpus\":%d,\"csr_cache\":\"%s\",\"image_cache\":\"%s\",\"exchange\":\"%s\",\"iters\":%d,\"nnz\":%.0f,\"rows\":%lld,\"steps_per_chunk\":%d,\"chunks\":%lld,\"rows_cut\":%lld,"
           "\"preprocess_s\":%.6g,\"spmv_compute_s\":%.6g,\"spmv_compute_median_s\":%.6g,\"spmv_with_gather_s\":%.6g,\"spmv_with_gather_median_s\":%.6g,\"gflops_2nnz\":%.6g,\"gbs_alg\":%.6g,"
           "\"frac_of_8TBs_per_gpu\":%.4f,\"wrong\":%lld}\n",
           G, !use_cache ? "off" : csr_hit ? "hit" : "miss", image_cache, G == 1 ? "none" : cvr_multi_uses_rccl(M) ? "rccl" : "copies", niters, nnz_true, (long long)m.ref_numRows, S0, (long long)chunks, (long long)cut, pre_s, t_compute,
           tm.median_s, t_total, tm.step_median_s, 2.0 * nnz_true / t_total / 1e9, balg / t_total / 1e9, balg / t_compute / (8e12 * G), (long long)wrong);
    cvr_destroy_multi(M);

    // CVR_POWER=<iterations>: the iterative caller (one GPU, square matrix): x <- A x / ||A x||
    const char *penv = getenv("CVR_POWER");
    if (penv && atoi(penv) > 0 && G == 1 && m.nrows == m.ncols) {
        cvr_handle *H = nullptr;
        opt.device = devs[0];
        CVR_OKAY(cvr_create(&H, &view, &opt));
        CVR_OKAY(cvr_preprocess(H, 0, nullptr));
        std::vector<double> ones((size_t)m.ncols, 1.0);
        HIP_OK(hipSetDevice(devs[0]));
        HIP_OK(hipMemcpy(cvr_x_device(H), ones.data(), sizeof(double) * (size_t)m.ncols, hipMemcpyHostToDevice));
        double lambda = 0, sec = 0;
        CVR_OKAY(cvr_power_iteration(H, nullptr, nullptr, atoi(penv), cvr_x_device(H), &lambda, &sec, cvr_stream(H)));
        printf("{\"power_iterations\":%d,\"rayleigh_quotient\":%.15g,\"seconds_per_iteration\":%.6g}\n", atoi(penv), lambda, sec);
        cvr_destroy(H);
    }
    cvr_mm_free(&m);
    return 0;
}

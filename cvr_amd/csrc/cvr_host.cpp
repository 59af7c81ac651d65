// cvr_host.cpp -- the small host-side pieces of the reference program that sit beside the hot path:
// fill (spmv.cpp:556-563), the CSR self-check loop (spmv.cpp:1843-1850) and the verdict (spmv.cpp:1916-1938).
#include <cmath>
#include <cstdint>

#include "../../include/cvr_amd.h"

extern "C" void cvr_fill_x(double *x, int64_t n, int mode)
{
    if (!x) return;
    if (mode == 0) {
        for (int64_t j = 0; j < n; j++) x[j] = 1.0;   // fill, spmv.cpp:556-563
        return;
    }
    for (int64_t j = 0; j < n; j++) {                 // splitmix64(0xC0FFEE, j) -> [-1, 1)  
        uint64_t z = 0xC0FFEEull + ((uint64_t)j + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        x[j] = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
}

extern "C" void cvr_csr_spmv_host(int64_t nrows, const int64_t *rp, const int32_t *ci, const double *va, const double *x,
                                  double *y, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < nrows; i++) {             // spmv.cpp:1843-1850: j ascending, plain sum
        double sum = 0;
        for (int64_t j = rp[i]; j < rp[i + 1]; j++) sum += va[j] * x[ci[j]];
        y[i] = sum;
    }
}

extern "C" int64_t cvr_verdict(const double *y, const double *yref, int64_t n)
{
    int64_t wrong = 0;
    for (int64_t i = 0; i < n; i++) {                 // spmv.cpp:1920-1929
        const double d = std::fabs(y[i] - yref[i]);
        if (d * d > 0.000001) wrong++;
    }
    return wrong;
}

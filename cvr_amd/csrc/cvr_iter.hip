// cvr_iter.hip -- vector kernels of the iterative caller (y <- A x chained, power iteration):
// dot products with a fixed reduction tree (bitwise reproducible from run to run), the normalisation x <- y / ||y||, and
// the un-padding of an all-gathered y.  The reference has no such loop (its Ntimes loop recomputes the same y,
// spmv.cpp:1024); this is the consumer a web-graph SpMV is built for.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cvr_kernels.h"

namespace cvr {
namespace {

constexpr int kDotBlocks = 1024, kDotThreads = 256;

// partial[b] = sum over the block's strided share of a[i] * b[i], accumulated in fp64 in a fixed order
template <typename T>
__global__ __launch_bounds__(kDotThreads) void dot_partial_kernel(const T *__restrict__ a, const T *__restrict__ b, long long n,
                                                                  double *__restrict__ partial)
{
    __shared__ double wsum[kDotThreads / 64];
    double acc = 0;
    for (long long i = (long long)blockIdx.x * kDotThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kDotThreads)
        acc += (double)a[i] * (double)b[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < kDotThreads / 64; w++) s += wsum[w];
        partial[blockIdx.x] = s;
    }
}

// out[0] = sum of the partials, one wavefront, fixed order
__global__ __launch_bounds__(64) void dot_final_kernel(const double *__restrict__ partial, int n, double *__restrict__ out)
{
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 64) acc += partial[(size_t)blockIdx.x * n + i];     // block b reduces the b-th set of partials
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// x[i] = y[i] / sqrt(norm2[0])   (norm2 = 0: x = 0)
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T *__restrict__ x, const T *__restrict__ y, const double *__restrict__ norm2, long long n)
{
    const double nn = norm2[0];
    const double inv = nn > 0 ? 1.0 / sqrt(nn) : 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] = (T)((double)y[i] * inv);
}

// One power-iteration step's vector work in ONE pass over x and y = A x: the partial sums of x . y, y . y and x . x of this step, and
// x <- y * inv with inv = 1 / ||y of the step before|| (each workgroup sums that step's partials itself, in dot_final_kernel's
// order: the same bits in every workgroup; the first step has none: inv = 1).  Normalising with the norm of the step before keeps ||x|| between 1/lambda and
// lambda (a cycle of six steps) without waiting for this step's reduction; the Rayleigh quotient x . y / x . x does not
// depend on the scale, and cvr_power_iteration normalises the last iterate exactly.
// PADDED: y is the all-gathered vector of equal-count slices (shard p's rows at y[p * max_rows ...]); the element order of the sums is
// the dense form's, so the sharded loop gives the same bits as the one-GPU loop without an un-padding pass per step.
template <typename T, bool PADDED>
__global__ __launch_bounds__(kDotThreads) void power_step_kernel(T *__restrict__ x, const T *__restrict__ y, long long n,
                                                                 const double *__restrict__ prev, double *__restrict__ out, IterBounds bd, int nparts,
                                                                 long long max_rows)
{
    __shared__ double wsum[3][kDotThreads / 64];
    __shared__ double pyy;
    if (prev && threadIdx.x < 64) {
        double acc = 0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) acc += prev[(size_t)gridDim.x + i];      // the y . y partials of the step before
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (threadIdx.x == 0) pyy = acc;
    }
    __syncthreads();
    const double inv = prev ? (pyy > 0 ? 1.0 / sqrt(pyy) : 0.0) : 1.0;
    double axy = 0, ayy = 0, axx = 0;
    for (long long i = (long long)blockIdx.x * kDotThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kDotThreads) {
        long long j = i;
        if constexpr (PADDED) {
            int lo = 0, hi = nparts - 1;                   // the shard of row i: last p with b[p] <= i
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (bd.b[mid] <= i) lo = mid; else hi = mid - 1; }
            j = (long long)lo * max_rows + (i - bd.b[lo]);
        }
        const double xv = (double)x[i], yv = (double)y[j];
        axy += xv * yv;
        ayy += yv * yv;
        axx += xv * xv;
        x[i] = (T)(yv * inv);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { axy += __shfl_xor(axy, o); ayy += __shfl_xor(ayy, o); axx += __shfl_xor(axx, o); }
    if ((threadIdx.x & 63u) == 0) { wsum[0][threadIdx.x >> 6] = axy; wsum[1][threadIdx.x >> 6] = ayy; wsum[2][threadIdx.x >> 6] = axx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int w = 0; w < kDotThreads / 64; w++) { s0 += wsum[0][w]; s1 += wsum[1][w]; s2 += wsum[2][w]; }
        out[blockIdx.x] = s0;
        out[gridDim.x + blockIdx.x] = s1;
        out[2 * gridDim.x + blockIdx.x] = s2;
    }
}

// dense[bounds[p] + i] = padded[p * max_rows + i]: the rows of an equal-count all-gather back in row order
template <typename T>
__global__ __launch_bounds__(256) void unpad_kernel(T *__restrict__ dense, const T *__restrict__ padded, IterBounds bd, long long max_rows)
{
    const int p = blockIdx.y;
    const long long r0 = bd.b[p], n = bd.b[p + 1] - r0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dense[r0 + i] = padded[(long long)p * max_rows + i];
}

}  // namespace

int dot_partials() { return 2 * kDotBlocks; }
int power_partials() { return 3 * kDotBlocks; }        // one step's partial sums (x . y, y . y, x . x)

hipError_t launch_power_step(void *x, const void *y, int64_t n, bool f32, const double *prev, double *out, hipStream_t st, const IterBounds *padded,
                             int nparts, int64_t max_rows)
{
    const IterBounds none{};
#define CVR_STEP(T, P) hipLaunchKernelGGL((power_step_kernel<T, P>), dim3(kDotBlocks), dim3(kDotThreads), 0, st, static_cast<T *>(x), static_cast<const T *>(y), \
                                          (long long)n, prev, out, padded ? *padded : none, nparts, (long long)max_rows)
    if (f32) { if (padded) CVR_STEP(float, true); else CVR_STEP(float, false); }
    else     { if (padded) CVR_STEP(double, true); else CVR_STEP(double, false); }
#undef CVR_STEP
    return hipGetLastError();
}

// cells[0 .. 2] = the sums of a step's three sets of partials
hipError_t launch_power_sums(const double *partial, double *cells, hipStream_t st)
{
    hipLaunchKernelGGL(dot_final_kernel, dim3(3), dim3(64), 0, st, partial, kDotBlocks, cells);
    return hipGetLastError();
}

hipError_t launch_dot(const void *a, const void *b, int64_t n, bool f32, double *partial, double *out, hipStream_t st)
{
    if (f32) hipLaunchKernelGGL(dot_partial_kernel<float>, dim3(kDotBlocks), dim3(kDotThreads), 0, st, static_cast<const float *>(a), static_cast<const float *>(b), (long long)n, partial);
    else hipLaunchKernelGGL(dot_partial_kernel<double>, dim3(kDotBlocks), dim3(kDotThreads), 0, st, static_cast<const double *>(a), static_cast<const double *>(b), (long long)n, partial);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(64), 0, st, partial, kDotBlocks, out);
    return hipGetLastError();
}

hipError_t launch_scale(void *x, const void *y, const double *norm2, int64_t n, bool f32, hipStream_t st)
{
    const uint32_t blocks = (uint32_t)std::min<int64_t>(2048, (n + 255) / 256 > 0 ? (n + 255) / 256 : 1);
    if (f32) hipLaunchKernelGGL(scale_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<float *>(x), static_cast<const float *>(y), norm2, (long long)n);
    else hipLaunchKernelGGL(scale_kernel<double>, dim3(blocks), dim3(256), 0, st, static_cast<double *>(x), static_cast<const double *>(y), norm2, (long long)n);
    return hipGetLastError();
}

hipError_t launch_unpad(void *dense, const void *padded, const IterBounds &bd, int nparts, int64_t max_rows, bool f32, hipStream_t st)
{
    const dim3 grid(256, (uint32_t)nparts);
    if (f32) hipLaunchKernelGGL(unpad_kernel<float>, grid, dim3(256), 0, st, static_cast<float *>(dense), static_cast<const float *>(padded), bd, (long long)max_rows);
    else hipLaunchKernelGGL(unpad_kernel<double>, grid, dim3(256), 0, st, static_cast<double *>(dense), static_cast<const double *>(padded), bd, (long long)max_rows);
    return hipGetLastError();
}

}  // namespace cvr

// tools/ubench/gather_rate.hip -- what the chip sustains on scattered 8-byte gathers (one 128-byte line fill each): requests per
// second by table size (L2-resident, Infinity-Cache-resident), wavefronts per CU and independent gathers in flight per lane.
// The yardstick for the x gather of the SpMV kernels.  hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

template <int U, typename V>
__global__ __launch_bounds__(256) void gather_kernel(const V *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t n_per_thread, double *out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (uint32_t i = 0; i < n_per_thread; i += U) {
        uint32_t j[U];
#pragma unroll
        for (int u = 0; u < U; u++) j[u] = idx[(size_t)(i + u) * nt + t];        // coalesced index stream
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = table[j[u]];
#pragma unroll
        for (int u = 0; u < U; u++) acc += (double)v[u];
    }
    if (acc == 12345.678) out[t] = acc;
}

int main(int argc, char **argv)
{
    const int iters = 5;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double *out; hipMalloc(&out, 8 << 20);
    printf("# table_MB  waves/CU  in_flight/lane  elem_B   Mreq/launch   us    Greq/s   req/clk/CU(2.1GHz)\n");
    for (double mb : {1.0, 2.0, 4.0, 7.3, 38.8, 268.0}) {
        const size_t n = (size_t)(mb * 1e6 / 8);
        double *table; hipMalloc(&table, n * 8); hipMemset(table, 0, n * 8);
        for (int wpc : {8, 16, 32}) {
            const int blocks = 256 * wpc / 4;
            const size_t nt = (size_t)blocks * 256;
            const uint32_t per = 64;
            std::vector<uint32_t> h(nt * per);
            uint64_t s = 88172645463325252ull;
            for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % n); }
            uint32_t *idx; hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            auto run = [&](auto kern, int U, int eb) {
                for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (const double *)table, idx, per, out);
                hipEventRecord(e0);
                for (int w = 0; w < iters; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (const double *)table, idx, per, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / iters, req = (double)nt * per;
                printf("%8.1f  %8d  %8d  %6d  %10.2f  %8.1f  %7.1f  %6.3f\n", mb, wpc, U, eb, req / 1e6, us, req / us / 1e3, req / us / 1e-6 / 256 / 2.1e9);
            };
            run(gather_kernel<1, double>, 1, 8);
            run(gather_kernel<2, double>, 2, 8);
            run(gather_kernel<4, double>, 4, 8);
            run(gather_kernel<8, double>, 8, 8);
            hipFree(idx);
        }
        hipFree(table);
    }
    return 0;
}

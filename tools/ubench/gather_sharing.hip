// tools/ubench/gather_sharing.hip -- what a CU's vector L1 charges for a gather instruction by HOW its 64 lanes share 128-byte lines:
// k consecutive lanes read (different words of) one line, the 64 / k lines of an instruction are random lines of the table.  Separates
// "one L2 request per distinct line" from "one L1 tag look-up per lane": the question behind every column-sorted layout (round 4).
//   hipcc --offload-arch=gfx950 -O3 gather_sharing.hip -o gather_sharing
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int U, typename V>
__global__ __launch_bounds__(256) void gather_kernel(const V *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t n_per_thread, double *out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (uint32_t i = 0; i < n_per_thread; i += U) {
        uint32_t j[U];
#pragma unroll
        for (int u = 0; u < U; u++) j[u] = idx[(size_t)(i + u) * nt + t];
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = table[j[u]];
#pragma unroll
        for (int u = 0; u < U; u++) acc += (double)v[u];
    }
    if (acc == 12345.678) out[t] = acc;
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <typename V>
static void sweep(const char *name, double table_mb, int wpc)
{
    const int iters = 5;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double *out; CK(hipMalloc(&out, 8 << 20));
    const size_t n = (size_t)(table_mb * 1e6 / sizeof(V)), per_line = 128 / sizeof(V), nlines = n / per_line;
    V *table; CK(hipMalloc(&table, n * sizeof(V))); CK(hipMemset(table, 0, n * sizeof(V)));
    const int blocks = 256 * wpc / 4;
    const size_t nt = (size_t)blocks * 256;
    const uint32_t per = 64;
    // pattern: k lanes per line; mode 0: k consecutive lanes, distinct random words; mode 1: lanes i, i + 64/k, ... (strided lanes share); mode 2: same word (broadcast)
    for (int mode = 0; mode < 3; mode++)
        for (int k : {1, 2, 4, 8, 16, 64}) {
            if (mode > 0 && k == 1) continue;
            std::vector<uint32_t> h(nt * per);
            for (uint32_t i = 0; i < per; i++)
                for (size_t w = 0; w < nt / 64; w++) {
                    uint32_t lines[64];
                    for (int q = 0; q < 64 / k; q++) lines[q] = (uint32_t)(rnd() % nlines);
                    for (int l = 0; l < 64; l++) {
                        const int q = mode == 1 ? l % (64 / k) : l / k;
                        const uint32_t word = mode == 2 ? 0u : (uint32_t)(rnd() % per_line);
                        h[(size_t)i * nt + w * 64 + l] = lines[q] * (uint32_t)per_line + word;
                    }
                }
            uint32_t *idx; CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            for (int w = 0; w < 2; w++) hipLaunchKernelGGL((gather_kernel<4, V>), dim3(blocks), dim3(256), 0, 0, (const V *)table, idx, per, out);
            CK(hipEventRecord(e0));
            for (int w = 0; w < iters; w++) hipLaunchKernelGGL((gather_kernel<4, V>), dim3(blocks), dim3(256), 0, 0, (const V *)table, idx, per, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters, lanes = (double)nt * per;
            printf("%-4s table %7.3f MB  waves/CU %2d  mode %d  lanes/line %2d : %8.1f us  %7.1f G lanes/s  %6.3f lanes/clk/CU  %7.1f G lines/s\n", name, table_mb, wpc, mode, k, us, lanes / us / 1e3,
                   lanes / us / 1e-6 / 256 / 2.1e9, lanes / k / us / 1e3);
            fflush(stdout);
            CK(hipFree(idx));
        }
    CK(hipFree(table)); CK(hipFree(out));
}

int main()
{
    for (double mb : {0.016, 2.0, 38.8}) {
        sweep<double>("f64", mb, 16);
        sweep<float>("f32", mb, 16);
    }
    sweep<double>("f64", 2.0, 4);
    return 0;
}

// tools/ubench/smem_rate.hip -- how many scalar loads (s_load_dword, one per 64-byte sector of a buffer far larger than the caches) can a CU
// keep going?  Round 5: spmv_ilv_kernel's helper wavefronts prefetch the matrix stream into the L2 through the scalar data cache, a
// path beside the vector L1's in-order miss queue; this measures that path's rate per CU by wavefronts per CU and loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 smem_rate.hip -o smem_rate && ./smem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int INFLIGHT>
__global__ __launch_bounds__(1024) void smem_kernel(const uint8_t *__restrict__ buf, size_t bytes_per_wave, uint32_t stride, uint32_t *sink)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t b0 = reinterpret_cast<uint64_t>(buf + (size_t)wave * bytes_per_wave);
    const uint64_t base = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b0) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32)) << 32);
    uint32_t acc = 0;
    for (uint32_t off = 0; off + INFLIGHT * stride <= bytes_per_wave; off += INFLIGHT * stride) {
        uint32_t j[15];
#pragma unroll
        for (int i = 0; i < INFLIGHT; i++) { const uint32_t o = off + i * stride; asm volatile("s_load_dword %0, %1, %2" : "=&s"(j[i]) : "s"(base), "s"(o) : "memory"); }
#pragma unroll
        for (int i = INFLIGHT; i < 15; i++) j[i] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(j[0]), "+s"(j[1]), "+s"(j[2]), "+s"(j[3]), "+s"(j[4]), "+s"(j[5]), "+s"(j[6]), "+s"(j[7]), "+s"(j[8]), "+s"(j[9]), "+s"(j[10]), "+s"(j[11]), "+s"(j[12]), "+s"(j[13]), "+s"(j[14]) :: "memory");
        acc += j[0];
    }
    if (acc == 0x12345678u) *sink = acc;
}

// the same bytes through the vector path, 16 B per lane, for comparison (bytes_per_wave per wavefront)
__global__ __launch_bounds__(1024) void vmem_kernel(const uint8_t *__restrict__ buf, size_t bytes_per_wave, uint32_t *sink)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint4 *p = reinterpret_cast<const uint4 *>(buf + (size_t)wave * bytes_per_wave);
    uint32_t acc = 0;
    for (size_t i = lane; i < bytes_per_wave / 16; i += 64 * 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = i + 64 * u < bytes_per_wave / 16 ? p[i + 64 * u] : uint4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t total = (size_t)3 << 30;          // 3 GiB: beyond the 256-MiB Infinity Cache
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# %d CUs; scalar loads of one dword per 64-byte sector (stride 64) or per 128-byte line (stride 128), fresh bytes of a 3-GiB buffer (HBM)\n", cus);
    printf("# waves/CU  in flight/wave  stride   loads/us/CU   GB/s (sectors touched x 64 B)\n");
    size_t cursor = 0;
    for (int wpc : {1, 2, 4, 8, 12, 16}) {
        for (int infl : {4, 8, 15}) {
            for (uint32_t stride : {64u, 128u}) {
                const int waves = cus * wpc;
                size_t per_wave = ((size_t)8 << 20) / wpc;          // 8 MiB per CU and run (256 CUs: 2 GiB of the 3)
                per_wave = per_wave / (15 * 128) * (15 * 128);
                if (cursor + (size_t)waves * per_wave > total) cursor = 0;
                const dim3 grid(cus), block(64 * wpc);
                CK(hipEventRecord(e0));
                if (infl == 4) hipLaunchKernelGGL(smem_kernel<4>, grid, block, 0, 0, buf + cursor, per_wave, stride, sink);
                else if (infl == 8) hipLaunchKernelGGL(smem_kernel<8>, grid, block, 0, 0, buf + cursor, per_wave, stride, sink);
                else hipLaunchKernelGGL(smem_kernel<15>, grid, block, 0, 0, buf + cursor, per_wave, stride, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                cursor += (size_t)waves * per_wave;
                const double loads = (double)waves * (double)(per_wave / stride);
                printf("  %6d %12d %10u %12.1f %10.0f\n", wpc, infl, stride, loads / (ms * 1e3) / cus, loads * 64.0 / (ms * 1e-3) / 1e9);
            }
        }
    }
    printf("# vector path, 16 B per lane, 4 loads in flight per lane\n");
    for (int wpc : {4, 8, 16}) {
        const int waves = cus * wpc;
        size_t per_wave = (((size_t)8 << 20) / wpc) & ~(size_t)4095;
        if (cursor + (size_t)waves * per_wave > total) cursor = 0;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(vmem_kernel, dim3(cus), dim3(64 * wpc), 0, 0, buf + cursor, per_wave, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        cursor += (size_t)waves * per_wave;
        printf("  %6d waves/CU: %8.0f GB/s, %8.1f lines/us/CU\n", wpc, (double)waves * per_wave / (ms * 1e-3) / 1e9, (double)waves * per_wave / 128.0 / (ms * 1e3) / cus);
    }
    return 0;
}

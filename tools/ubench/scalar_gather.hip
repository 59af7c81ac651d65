// scalar_gather.hip -- microbenchmark: random 8-byte gathers from a 7.3-MB vector through the vector memory path
// (buffer/global loads, CU L1 -> L2, 128-B fills) against the scalar path (s_load_dwordx2, scalar cache -> L2, 64-B
// fills), and both at once.  Question: is the scalar cache's miss path a second, independent gather pipe?
// build: hipcc -O3 --offload-arch=gfx950 -o scalar_gather scalar_gather.hip ; run: ./scalar_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

constexpr int kPerLane = 56 * 4 / 4;   // gathers per lane per wave = steps of a chunk (S = 56)

// vector path: lane-private indices, 4 independent gathers in flight per lane per iteration
__global__ __launch_bounds__(64) void vec_gather(const uint32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out, int per_lane)
{
    const uint32_t *p = idx + (size_t)blockIdx.x * per_lane * 64 + threadIdx.x * 4;
    double acc = 0;
    for (int i = 0; i < per_lane; i += 4) {
        const uint4 c = *reinterpret_cast<const uint4 *>(p + (size_t)i * 64);
        acc += x[c.x] + x[c.y] + x[c.z] + x[c.w];
    }
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}

// scalar path: the wave walks its lanes' indices with readlane and loads through the scalar cache, `SB` loads per batch
template <int SB>
__global__ __launch_bounds__(64) void sc_gather(const uint32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out, int per_lane)
{
    const uint32_t *p = idx + (size_t)blockIdx.x * per_lane * 64 + threadIdx.x * 4;
    double acc = 0;
    for (int i = 0; i < per_lane; i += 4) {
        const uint4 c = *reinterpret_cast<const uint4 *>(p + (size_t)i * 64);
        const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int l0 = 0; l0 < 64; l0 += SB) {
                double v[SB];
#pragma unroll
                for (int l = 0; l < SB; l++) {
                    const uint32_t s = __builtin_amdgcn_readlane(cc[j], l0 + l);
                    v[l] = x[s];                                   // uniform address: s_load_dwordx2
                }
#pragma unroll
                for (int l = 0; l < SB; l++) acc += v[l];          // (uniform sum; a real kernel would v_writelane)
            }
        }
    }
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}

// both: of every 4 gathers of a lane, VEC go through the vector path and 4 - VEC through the scalar path
template <int VEC, int SB>
__global__ __launch_bounds__(64) void mix_gather(const uint32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out, int per_lane)
{
    const uint32_t *p = idx + (size_t)blockIdx.x * per_lane * 64 + threadIdx.x * 4;
    double acc = 0;
    for (int i = 0; i < per_lane; i += 4) {
        const uint4 c = *reinterpret_cast<const uint4 *>(p + (size_t)i * 64);
        const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
        double xv[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < VEC; j++) xv[j] = x[cc[j]];
#pragma unroll
        for (int j = VEC; j < 4; j++) {
#pragma unroll
            for (int l0 = 0; l0 < 64; l0 += SB) {
                double v[SB];
#pragma unroll
                for (int l = 0; l < SB; l++) v[l] = x[__builtin_amdgcn_readlane(cc[j], l0 + l)];
#pragma unroll
                for (int l = 0; l < SB; l++) acc += v[l];
            }
        }
        acc += xv[0] + xv[1] + xv[2] + xv[3];
    }
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    // default: 4x the web-Google launch (21.5 M gathers); argv: waves, gathers per lane (multiple of 4)
    const size_t nx = 916428, nwaves = argc > 1 ? (size_t)atol(argv[1]) : 1497, per_lane = argc > 2 ? (size_t)atol(argv[2]) : 224, n = nwaves * per_lane * 64;
    printf("%zu waves x %zu gathers per lane\n", nwaves, per_lane);
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % nx); }
    if (argc > 3) {     // a real column stream (u32, values < nx), repeated / truncated to n entries
        FILE *f = fopen(argv[3], "rb");
        if (!f) { printf("cannot open %s\n", argv[3]); return 1; }
        std::vector<uint32_t> t(n);
        const size_t got = fread(t.data(), 4, n, f);
        fclose(f);
        if (got == 0) { printf("empty index file\n"); return 1; }
        for (size_t i = 0; i < n; i++) h[i] = t[i % got] < nx ? t[i % got] : 0;
        printf("column stream from %s (%zu entries)\n", argv[3], got);
    }
    std::vector<double> hx(nx, 1.0);
    uint32_t *d_idx; double *d_x, *d_out;
    CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&d_x, nx * 8)); CK(hipMalloc(&d_out, nwaves * 64 * 8));
    CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_x, hx.data(), nx * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; i++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double chk; CK(hipMemcpy(&chk, d_out, 8, hipMemcpyDeviceToHost));
        printf("%-34s %8.2f us per launch  %7.1f G gathers/s  (check %.0f)\n", name, ms * 1e3 / 50, n / (ms * 1e-3 / 50) / 1e9, chk);
    };
    time_it("vector path", [&] { hipLaunchKernelGGL(vec_gather, dim3(nwaves), dim3(64), 0, 0, d_idx, d_x, d_out, (int)per_lane); });
    time_it("scalar path, 8 per batch", [&] { hipLaunchKernelGGL(sc_gather<8>, dim3(nwaves), dim3(64), 0, 0, d_idx, d_x, d_out, (int)per_lane); });
    time_it("scalar path, 16 per batch", [&] { hipLaunchKernelGGL(sc_gather<16>, dim3(nwaves), dim3(64), 0, 0, d_idx, d_x, d_out, (int)per_lane); });
    time_it("3 vector : 1 scalar (8/batch)", [&] { hipLaunchKernelGGL((mix_gather<3, 8>), dim3(nwaves), dim3(64), 0, 0, d_idx, d_x, d_out, (int)per_lane); });
    time_it("2 vector : 2 scalar (8/batch)", [&] { hipLaunchKernelGGL((mix_gather<2, 8>), dim3(nwaves), dim3(64), 0, 0, d_idx, d_x, d_out, (int)per_lane); });
    time_it("3 vector : 1 scalar (16/batch)", [&] { hipLaunchKernelGGL((mix_gather<3, 16>), dim3(nwaves), dim3(64), 0, 0, d_idx, d_x, d_out, (int)per_lane); });
    return 0;
}

// tools/ubench/long_rows.hip -- prototype of the LONG-ROW kernel (round 4): rows of at least L non-zeros leave the CVR image; the 64 lanes
// of a wavefront take 64 CONSECUTIVE (column-sorted) non-zeros of ONE row per step -- the gather instruction reads neighbours, lanes
// share lines -- keep their running sums in registers and fold them with a wave butterfly; a row longer than `gmax` groups is cut into
// pieces, summed in order by a second kernel.  Measures what the long rows of a power-law matrix cost this way before the library is
// changed for it.   hipcc --offload-arch=gfx950 -O3 -fopenmp long_rows.hip -o long_rows ;  long_rows csr.bin f32|f64 L gmax iters
#include <hip/hip_runtime.h>
#include <omp.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef double   f64x2 __attribute__((ext_vector_type(2)));

struct Piece { uint64_t goff; uint32_t ngroups, slot; };      // groups [goff, goff + ngroups) of the stream; partial sum -> part[slot]

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000); }

template <typename T> __global__ __launch_bounds__(256) void long_kernel(const uint8_t *__restrict__ stream, const Piece *__restrict__ pieces, uint32_t npieces, const T *__restrict__ x,
                                                                          uint32_t xbytes, T *__restrict__ part)
{
    constexpr uint32_t GB = 1024u + (sizeof(T) == 8 ? 2048u : 1024u);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (k >= npieces) return;
    const Piece p = pieces[k];
    const uint8_t *base = stream + p.goff * GB;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, xbytes);
    T acc = 0;
    for (uint32_t g = 0; g < p.ngroups; g++) {
        const uint8_t *grp = base + (size_t)g * GB;
        const u32x4 c = *reinterpret_cast<const u32x4 *>(grp + lane * 16);
        T v[4];
        if constexpr (sizeof(T) == 8) {
            const f64x2 lo = *reinterpret_cast<const f64x2 *>(grp + 1024 + lane * 16), hi = *reinterpret_cast<const f64x2 *>(grp + 2048 + lane * 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
        } else {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(grp + 1024 + lane * 16);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
        const uint32_t col[4] = {c.x, c.y, c.z, c.w};
        T xv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (sizeof(T) == 8) xv[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, col[j] * 8u, 0, 0));
            else xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, col[j] * 4u, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc = sizeof(T) == 8 ? (T)__builtin_fma((double)v[j], (double)xv[j], (double)acc) : (T)__builtin_fmaf((float)v[j], (float)xv[j], (float)acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) part[p.slot] = acc;
}

// y[row] = its pieces' sums in order
template <typename T> __global__ void sum_kernel(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ first, uint32_t nrows_long, const T *__restrict__ part, T *__restrict__ y)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows_long) return;
    T s = 0;
    for (uint32_t q = first[i]; q < first[i + 1]; q++) s += part[q];
    y[rows[i]] = s;
}

template <typename T> int run(const char *path, int64_t L, uint32_t gmax, int iters)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    int64_t h[3];
    if (fread(h, 8, 3, f) != 3) return 1;
    const int64_t nrows = h[0], ncols = h[1], nnz = h[2];
    std::vector<int64_t> rp(nrows + 1); std::vector<int32_t> ci(nnz); std::vector<T> va(nnz);
    if (fread(rp.data(), 8, nrows + 1, f) != (size_t)nrows + 1 || fread(ci.data(), 4, nnz, f) != (size_t)nnz || fread(va.data(), sizeof(T), nnz, f) != (size_t)nnz) return 1;
    fclose(f);
    constexpr uint32_t GB = 1024u + (sizeof(T) == 8 ? 2048u : 1024u);
    std::vector<uint32_t> rows, first;
    std::vector<Piece> pieces;
    uint64_t ngroups = 0; int64_t nnz_long = 0;
    for (int64_t r = 0; r < nrows; r++) {
        const int64_t n = rp[r + 1] - rp[r];
        if (n < L) continue;
        rows.push_back((uint32_t)r); first.push_back((uint32_t)pieces.size());
        const uint32_t G = (uint32_t)((n + 255) / 256);
        for (uint32_t g0 = 0; g0 < G; g0 += gmax) { pieces.push_back({ngroups + g0, std::min(gmax, G - g0), (uint32_t)pieces.size()}); }
        ngroups += G; nnz_long += n;
    }
    first.push_back((uint32_t)pieces.size());
    std::vector<uint8_t> stream((size_t)(ngroups + 1) * GB, 0);
    {
        uint64_t g = 0;
        std::vector<uint64_t> gstart(rows.size());
        for (size_t i = 0; i < rows.size(); i++) { gstart[i] = g; g += (rp[rows[i] + 1] - rp[rows[i]] + 255) / 256; }
#pragma omp parallel for schedule(dynamic, 64)
        for (size_t i = 0; i < rows.size(); i++) {
            const int64_t b = rp[rows[i]], n = rp[rows[i] + 1] - b;
            for (int64_t e = 0; e < (int64_t)((n + 255) / 256) * 256; e++) {
                uint8_t *grp = stream.data() + (gstart[i] + e / 256) * GB;
                const uint32_t j = (uint32_t)(e % 256) / 64, ln = (uint32_t)(e % 64);
                const uint32_t col = e < n ? (uint32_t)ci[b + e] : (uint32_t)ci[b + n - 1];
                const T v = e < n ? va[b + e] : T(0);
                reinterpret_cast<uint32_t *>(grp)[ln * 4 + j] = col;
                if (sizeof(T) == 8) reinterpret_cast<double *>(grp + 1024 + (j >= 2 ? 1024 : 0))[ln * 2 + (j & 1)] = (double)v;
                else reinterpret_cast<float *>(grp + 1024)[ln * 4 + j] = (float)v;
            }
        }
    }
    std::vector<T> x(ncols + 64);
    for (int64_t j = 0; j < ncols; j++) {
        uint64_t zz = 0xC0FFEEull + ((uint64_t)j + 1) * 0x9E3779B97F4A7C15ull;
        zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull; zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull; zz ^= zz >> 31;
        x[j] = (T)((double)(zz >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0);
    }
    uint8_t *d_stream; Piece *d_pieces; uint32_t *d_rows, *d_first; T *d_x, *d_part, *d_y;
    CK(hipMalloc(&d_stream, stream.size())); CK(hipMemcpy(d_stream, stream.data(), stream.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pieces, std::max<size_t>(pieces.size(), 1) * sizeof(Piece))); CK(hipMemcpy(d_pieces, pieces.data(), pieces.size() * sizeof(Piece), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_rows, std::max<size_t>(rows.size(), 1) * 4)); CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_first, first.size() * 4)); CK(hipMemcpy(d_first, first.data(), first.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, x.size() * sizeof(T))); CK(hipMemcpy(d_x, x.data(), x.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_part, std::max<size_t>(pieces.size(), 1) * sizeof(T))); CK(hipMalloc(&d_y, nrows * sizeof(T))); CK(hipMemset(d_y, 0, nrows * sizeof(T)));
    const uint32_t np = (uint32_t)pieces.size(), nl = (uint32_t)rows.size();
    auto launch = [&]() {
        hipLaunchKernelGGL(long_kernel<T>, dim3((np + 3) / 4), dim3(256), 0, 0, d_stream, d_pieces, np, d_x, (uint32_t)(ncols * sizeof(T)), d_part);
        hipLaunchKernelGGL(sum_kernel<T>, dim3((nl + 255) / 256), dim3(256), 0, 0, d_rows, d_first, nl, d_part, d_y);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<T> y(nrows);
    CK(hipMemcpy(y.data(), d_y, nrows * sizeof(T), hipMemcpyDeviceToHost));
    int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad)
    for (size_t i = 0; i < rows.size(); i++) {
        const int64_t r = rows[i];
        double s = 0, a = 0;
        for (int64_t j = rp[r]; j < rp[r + 1]; j++) { const double pv = (double)va[j] * (double)x[ci[j]]; s += pv; a += std::fabs(pv); }
        if (std::fabs((double)y[r] - s) > (sizeof(T) == 8 ? 1e-12 : 1e-5) * a + 1e-300) bad++;
    }
    printf("L %ld gmax %u: long rows %zu (%.2f %% of rows) nnz %ld (%.1f %% of nnz) pieces %u groups %lu (slots/nnz %.3f) stream %.1f MB | %.1f us  %.0f GB/s stream  wrong %ld\n", (long)L, gmax, rows.size(),
           100.0 * rows.size() / nrows, (long)nnz_long, 100.0 * nnz_long / nnz, np, (unsigned long)ngroups, ngroups * 256.0 / std::max<int64_t>(nnz_long, 1), ngroups * (double)GB / 1e6, ms * 1e3 / iters,
           ngroups * (double)GB / (ms * 1e-3 / iters) / 1e9, (long)bad);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s csr.bin f32|f64 L gmax iters\n", argv[0]); return 1; }
    return !strcmp(argv[2], "f32") ? run<float>(argv[1], atoll(argv[3]), (uint32_t)atoi(argv[4]), atoi(argv[5])) : run<double>(argv[1], atoll(argv[3]), (uint32_t)atoi(argv[4]), atoi(argv[5]));
}

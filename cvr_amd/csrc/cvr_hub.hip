// cvr_hub.hip -- hub columns: the H columns that hold the most non-zeros get their x values staged in LDS.
//
// A power-law matrix (R-MAT, web and social graphs) sends a large share of its gathers to a few thousand columns, and on
// this chip every gather that misses the 32-KB L1 is a 128-byte fill request: R-MAT scale 22 issues 0.92
// L1->L2 requests per non-zero although the 32 768 most popular of its 4 M columns hold half of them.  With a hub table
// the workgroup (several chunks) copies x[hub columns] -- compacted into one contiguous array by a tiny kernel before
// every SpMV -- into LDS with coalesced loads and serves those gathers by ds_read; the column word of such a slot holds
// the table index and bit 30.  Selection: columns by number of non-zeros, descending, ties by column index (stable
// radix sort), cut at the LDS budget; the reference has no counterpart (its gather is _mm512_i32logather_pd on the whole
// x, spmv.cpp:1227, with a software prefetch, spmv.cpp:1183-1190).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "cvr_kernels.h"

namespace cvr {
namespace {

// every stride-th non-zero is counted (a sample of at most ~8 M keeps the pass at a millisecond or two; the ranking of the
// popular columns does not need more)
__global__ __launch_bounds__(256) void hub_count_kernel(const int32_t *__restrict__ ci, long long n0, long long nsamp, long long stride, uint32_t *__restrict__ cnt)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < nsamp; t += (long long)gridDim.x * 256) atomicAdd(&cnt[ci[n0 + t * stride]], 1u);
}

__global__ __launch_bounds__(256) void hub_bitmap_kernel(const int32_t *__restrict__ hub_cols, uint32_t H, uint32_t *__restrict__ bitmap)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256) atomicOr(&bitmap[(uint32_t)hub_cols[i] >> 5], 1u << ((uint32_t)hub_cols[i] & 31u));
}

__global__ __launch_bounds__(256) void hub_iota_kernel(int32_t *__restrict__ v, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) v[i] = (int32_t)i;
}

// number of leading entries of the descending counts that are >= 2 (at most hmax), and the sum of the first that many
__global__ __launch_bounds__(1024) void hub_cut_kernel(const uint32_t *__restrict__ sorted_cnt, uint32_t ncols, uint32_t hmax,
                                                       unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t           hcut;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = hmax < ncols ? hmax : ncols;       // first position with count < 2
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted_cnt[mid] >= 2) lo = mid + 1; else hi = mid; }
        hcut = lo;
    }
    __syncthreads();
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < hcut; i += 1024) s += sorted_cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { out[0] = hcut; out[1] = part[0]; }
}

__global__ __launch_bounds__(256) void hub_index_kernel(const int32_t *__restrict__ hub_cols, uint32_t H, int32_t *__restrict__ hub_index)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256) hub_index[hub_cols[i]] = (int32_t)i;
}

template <typename T>
__global__ __launch_bounds__(256) void hub_gather_kernel(const T *__restrict__ x, const int32_t *__restrict__ hub_cols, uint32_t H, uint32_t Hpad,
                                                         T *__restrict__ xh)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < Hpad) xh[i] = i < H ? x[hub_cols[i]] : T(0);
}

}  // namespace

hipError_t select_hubs(const int32_t *ci, int64_t n0, int64_t n1, int64_t ncols, uint32_t hmax, HubSelection *out, hipStream_t st, bool full_order)
{
    *out = HubSelection{};
    if (n1 <= n0 || ncols <= 0 || hmax == 0) return hipSuccess;
    uint32_t *cnt = nullptr, *cnt_s = nullptr;
    int32_t  *col = nullptr, *col_s = nullptr;
    void     *tmp = nullptr, *arena = nullptr;
    unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
    size_t    tmp_bytes = 0;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t err) { (void)hipFree(arena); return err; };      // (one allocation for the pass's six buffers: every hipFree waits for the device)
#define HUB_TRY(x) do { e = (x); if (e != hipSuccess) return done(e); } while (0)
    const size_t nc = (size_t)ncols;
    HUB_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, cnt, cnt_s, col, col_s, (int)nc, 0, 32, st));      // (size query: no work)
    {
        auto         up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_cnt = 0, o_cnts = o_cnt + up(4 * nc), o_col = o_cnts + up(4 * nc), o_cols = o_col + up(4 * nc), o_out = o_cols + up(4 * nc), o_tmp = o_out + 256;
        HUB_TRY(hipMalloc(&arena, o_tmp + up(tmp_bytes ? tmp_bytes : 16)));
        uint8_t *a = static_cast<uint8_t *>(arena);
        cnt = reinterpret_cast<uint32_t *>(a + o_cnt); cnt_s = reinterpret_cast<uint32_t *>(a + o_cnts);
        col = reinterpret_cast<int32_t *>(a + o_col); col_s = reinterpret_cast<int32_t *>(a + o_cols);
        d_out = reinterpret_cast<unsigned long long *>(a + o_out); tmp = a + o_tmp;
    }
    HUB_TRY(hipMemsetAsync(cnt, 0, 4 * nc, st));
    const int64_t stride = hub_sample_stride(n1 - n0), nsamp = (n1 - n0 + stride - 1) / stride;
    const uint32_t blocks = (uint32_t)std::min<int64_t>(8192, (nsamp + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(hub_count_kernel, dim3(blocks), dim3(256), 0, st, ci, (long long)n0, (long long)nsamp, (long long)stride, cnt);
    hipLaunchKernelGGL(hub_iota_kernel, dim3((uint32_t)std::min<size_t>(4096, (nc + 255) / 256)), dim3(256), 0, st, col, (uint32_t)nc);
    HUB_TRY(hipGetLastError());
    HUB_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp, tmp_bytes, cnt, cnt_s, col, col_s, (int)nc, 0, 32, st));       // stable: ties by column
    hipLaunchKernelGGL(hub_cut_kernel, dim3(1), dim3(1024), 0, st, cnt_s, (uint32_t)nc, hmax, d_out);
    HUB_TRY(hipGetLastError());
    HUB_TRY(hipMemcpyAsync(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost, st));
    HUB_TRY(hipStreamSynchronize(st));
    out->H = (uint32_t)h_out[0];
    out->share = (double)h_out[1] / (double)nsamp;           // of the sampled non-zeros
    if (out->H > 0) {
        // full_order: every column gets its rank (the whole of x is re-ordered by popularity before every SpMV), not only the hubs
        const uint32_t nkeep = full_order ? (uint32_t)nc : out->H;
        out->order_n = full_order ? (uint32_t)nc : 0;
        HUB_TRY(hipMalloc(&out->hub_cols, 4 * (size_t)nkeep));
        HUB_TRY(hipMalloc(&out->hub_index, 4 * nc));
        HUB_TRY(hipMemcpyAsync(out->hub_cols, col_s, 4 * (size_t)nkeep, hipMemcpyDeviceToDevice, st));
        HUB_TRY(hipMemsetAsync(out->hub_index, 0xff, 4 * nc, st));
        HUB_TRY(hipMalloc(&out->hub_bitmap, 4 * ((nc + 31) / 32)));
        HUB_TRY(hipMemsetAsync(out->hub_bitmap, full_order ? 0xff : 0, 4 * ((nc + 31) / 32), st));
        hipLaunchKernelGGL(hub_index_kernel, dim3(std::min<uint32_t>(4096, (nkeep + 255) / 256)), dim3(256), 0, st, out->hub_cols, nkeep, out->hub_index);
        if (!full_order) hipLaunchKernelGGL(hub_bitmap_kernel, dim3((out->H + 255) / 256), dim3(256), 0, st, out->hub_cols, out->H, out->hub_bitmap);
        HUB_TRY(hipGetLastError());
        HUB_TRY(hipStreamSynchronize(st));
    }
#undef HUB_TRY
    return done(hipSuccess);
}

// per count value c < kShareBins: how many columns were counted c times; [kShareBins]: number and sum of the larger counts
constexpr uint32_t kShareBins = 4096;
// hist[kShareBins + 2 + b], b < kColBins: the sampled non-zeros whose column lies in the b-th of kColBins equal column ranges (what the panel rule
// weighs the panels' loads with: cvr_capi.hip, balanced_panel_count)
__global__ __launch_bounds__(256) void hub_share_kernel(const uint32_t *__restrict__ cnt, uint32_t ncols, unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t h[kShareBins];
    __shared__ uint32_t cb[kColBins];
    for (uint32_t i = threadIdx.x; i < kShareBins; i += 256) h[i] = 0;
    for (uint32_t i = threadIdx.x; i < kColBins; i += 256) cb[i] = 0;
    __syncthreads();
    unsigned long long nbig = 0, sbig = 0;
    const uint32_t     lane = threadIdx.x & 63u;
    for (uint32_t c0 = blockIdx.x * 256 + (threadIdx.x & ~63u); c0 < ncols; c0 += gridDim.x * 256) {          // (a wavefront's 64 columns at a time)
        const uint32_t c = c0 + lane;
        const uint32_t v = c < ncols ? cnt[c] : 0u;
        const uint32_t last = c0 + 63u < ncols ? c0 + 63u : ncols - 1u;
        const uint32_t b0 = (uint32_t)((unsigned long long)c0 * kColBins / ncols), b1 = (uint32_t)((unsigned long long)last * kColBins / ncols);
        if (b0 == b1) {          // all in one column range: one LDS atomic for the wavefront
            uint32_t sum = v;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (lane == 0 && sum) atomicAdd(&cb[b0], sum);
        } else if (v) atomicAdd(&cb[(uint32_t)((unsigned long long)c * kColBins / ncols)], v);
        if (v < 2) continue;
        if (v < kShareBins) atomicAdd(&h[v], 1u); else { nbig++; sbig += v; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kShareBins; i += 256) if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
    for (uint32_t i = threadIdx.x; i < kColBins; i += 256) if (cb[i]) atomicAdd(&hist[kShareBins + 2 + i], (unsigned long long)cb[i]);
    if (nbig) { atomicAdd(&hist[kShareBins], nbig); atomicAdd(&hist[kShareBins + 1], sbig); }
}

// The share of the (sampled) non-zeros that the hmax most popular columns hold -- select_hubs' `share` -- without ranking the columns:
// the columns' counts, then a histogram of the count VALUES, walked from the top until hmax columns are in.  cvr_create asks this of a
// matrix that is about to get column panels (are its popular columns worth hub tables?): the sort of all column counts it replaces was
// ~0.6 of the 1.0 ms of that question on the soc-LiveJournal1 shape.
hipError_t hub_share_device(const int32_t *ci, int64_t n0, int64_t n1, int64_t ncols, uint32_t hmax, double *share, hipStream_t st, Scratch lent, double *col_share)
{
    *share = 0;
    if (col_share) for (uint32_t b = 0; b < kColBins; b++) col_share[b] = 0;
    if (n1 <= n0 || ncols <= 0 || hmax == 0) return hipSuccess;
    void *arena = nullptr;
    const size_t nc = (size_t)ncols, o_hist = (4 * nc + 255) & ~(size_t)255, hist_bytes = sizeof(unsigned long long) * (kShareBins + 2 + kColBins);
    const bool   own = !(lent.p && lent.bytes >= o_hist + hist_bytes);          // (lent by cvr_create: the planner's scratch, idle until the split is done)
    hipError_t e = hipSuccess;
    if (own) e = hipMalloc(&arena, o_hist + hist_bytes); else arena = lent.p;
    if (e != hipSuccess) return e;
    uint32_t           *cnt = static_cast<uint32_t *>(arena);
    unsigned long long *hist = reinterpret_cast<unsigned long long *>(static_cast<uint8_t *>(arena) + o_hist);
    std::vector<unsigned long long> h(kShareBins + 2 + kColBins);
    e = hipMemsetAsync(arena, 0, o_hist + hist_bytes, st);
    const int64_t stride = hub_sample_stride(n1 - n0), nsamp = (n1 - n0 + stride - 1) / stride;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hub_count_kernel, dim3((uint32_t)std::min<int64_t>(8192, (nsamp + 256 * 8 - 1) / (256 * 8))), dim3(256), 0, st, ci, (long long)n0, (long long)nsamp, (long long)stride, cnt);
        hipLaunchKernelGGL(hub_share_kernel, dim3((uint32_t)std::min<size_t>(2048, (nc + 255) / 256)), dim3(256), 0, st, cnt, (uint32_t)nc, hist);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), hist, hist_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (own) (void)hipFree(arena);
    if (e != hipSuccess) return e;
    unsigned long long left = hmax, sum = 0;
    { const unsigned long long take = std::min(left, h[kShareBins]); sum += h[kShareBins] ? h[kShareBins + 1] * take / h[kShareBins] : 0; left -= take; }      // (more such columns than hmax cannot happen with a sample of 2^23)
    for (uint32_t v = kShareBins - 1; v >= 2 && left > 0; v--) { const unsigned long long take = std::min(left, h[v]); sum += take * v; left -= take; }
    *share = (double)sum / (double)nsamp;
    if (col_share) for (uint32_t b = 0; b < kColBins; b++) col_share[b] = (double)h[kShareBins + 2 + b] / (double)nsamp;
    return hipSuccess;
}

void free_hubs(HubSelection &s)
{
    (void)hipFree(s.hub_cols); (void)hipFree(s.hub_index); (void)hipFree(s.hub_bitmap);
    s = HubSelection{};
}

hipError_t launch_hub_gather(const DeviceImage &img, const void *x_ext, hipStream_t st)
{
    if (img.hub_n == 0) return hipSuccess;
    const uint32_t n = img.order_n ? img.order_n : img.hub_n;       // order_n: the whole of x, re-ordered; x_perm[ncols] = 0 is the pad slot
    const uint32_t Hpad = img.order_n ? (img.order_n + 1u + 3u) & ~3u : (img.hub_n + 3u) & ~3u;
    if (img.f32) hipLaunchKernelGGL(hub_gather_kernel<float>, dim3((Hpad + 255) / 256), dim3(256), 0, st, static_cast<const float *>(x_ext), img.hub_cols, n, Hpad, static_cast<float *>(img.hub_x));
    else hipLaunchKernelGGL(hub_gather_kernel<double>, dim3((Hpad + 255) / 256), dim3(256), 0, st, static_cast<const double *>(x_ext), img.hub_cols, n, Hpad, static_cast<double *>(img.hub_x));
    return hipGetLastError();
}

}  // namespace cvr

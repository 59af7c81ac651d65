// cvr_split.hip -- column-panel split of a CSR matrix that is already in device memory (cvr_csr_view.arrays_on_device):
// the device counterpart of split_panels_t in cvr_capi.hip.  Every non-zero gets its panel as a key; one stable radix
// sort pass (hipCUB) groups the non-zeros by panel and keeps their original order inside a panel, i.e. row after row,
// each row's entries as they were -- exactly what the host split produces by walking the rows.  Sub-row boundaries
// (a new row, or a new panel) are flagged and numbered with a prefix sum.  Nothing but the per-panel row pointers and
// row numbers (12 B per sub-row) has to travel to the host, where the planner needs them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "cvr_kernels.h"

namespace cvr {
namespace {

__global__ __launch_bounds__(256) void split_key_kernel(const int32_t *__restrict__ ci, long long nz0, long long n, long long width,
                                                        uint8_t *__restrict__ key, uint32_t *__restrict__ idx)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        key[t] = (uint8_t)(ci[nz0 + t] / width);
        idx[t] = (uint32_t)t;
    }
}

// off[p] = first position of the sorted keys with key >= p (p = 0 .. P)
__global__ void split_bounds_kernel(const uint8_t *__restrict__ key, long long n, int P, long long *__restrict__ off)
{
    const int p = threadIdx.x;
    if (p > P) return;
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((int)key[mid] < p) lo = mid + 1; else hi = mid;
    }
    off[p] = lo;
}

// the sorted non-zeros: column, value, row (binary search in row_ptr).  One workgroup per 256 elements (no grid-stride loop:
// the search is a chain of dependent loads).  What bounds it is not the search: a panel's elements are a ninth (1 / P) of every
// cache line of col_idx and vals, so the CSR is read P times over (LiveJournal shape: 1.8 ms; bracketing the search per
// wavefront or per run of 8 elements made it 3.1 - 3.8 ms)
template <typename T>
__global__ __launch_bounds__(256) void split_gather_kernel(const long long *__restrict__ rp, long long nrows, const int32_t *__restrict__ ci,
                                                           const T *__restrict__ va, long long nz0, long long n,
                                                           const uint32_t *__restrict__ idx, int32_t *__restrict__ ci_s, T *__restrict__ va_s,
                                                           uint32_t *__restrict__ row_s)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const long long j = nz0 + (long long)idx[t];
        long long lo = 0, hi = nrows;                       // last r with rp[r] <= j (then rp[r+1] > j)
        while (lo < hi) {
            const long long mid = (lo + hi + 1) >> 1;
            if (rp[mid] <= j) lo = mid; else hi = mid - 1;
        }
        ci_s[t] = ci[j];
        va_s[t] = va[j];
        row_s[t] = (uint32_t)lo;
    }
}

__global__ __launch_bounds__(256) void split_head_kernel(const uint8_t *__restrict__ key, const uint32_t *__restrict__ row_s, long long n,
                                                         uint32_t *__restrict__ head)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256)
        head[t] = (t == 0 || key[t] != key[t - 1] || row_s[t] != row_s[t - 1]) ? 1u : 0u;
}

// sub-row k starts at sorted position t: rows[k] = its row, rp[k] = t; rp[nsub] = n
__global__ __launch_bounds__(256) void split_emit_kernel(const uint32_t *__restrict__ head, const uint32_t *__restrict__ sidx,
                                                         const uint32_t *__restrict__ row_s, long long n, uint32_t *__restrict__ rows,
                                                         long long *__restrict__ rp_s, long long nsub)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256)
        if (head[t]) { rows[sidx[t]] = row_s[t]; rp_s[sidx[t]] = t; }
    if (blockIdx.x == 0 && threadIdx.x == 0) rp_s[nsub] = n;
}

// ---- the split as a stable partition (no sort, no search per non-zero) -------------------------------------------------
// The radix sort of (panel, position) pairs and the gather behind it read the CSR arrays P times over (a panel's elements are 1 / P of
// every cache line) and search row_ptr once per non-zero.  A partition by panel is a counting sort with P buckets: tiles of 2 048
// consecutive non-zeros count their elements per panel (part_count_kernel), one scan over the counts in (panel, tile) order gives
// every tile its place in every panel, and the tiles then move their elements there themselves (part_scatter_kernel): every thread
// owns eight consecutive non-zeros, counts them per panel in a column of its own in LDS, a scan in (panel, thread) order ranks them --
// the order inside a panel stays the CSR's, as the stable sort left it -- and the row of a thread's first element is one search, the
// others follow from row_ptr.  The CSR is read once (the column indices twice).
constexpr int kPartTile = 2048, kPartThreads = 256, kPartPer = kPartTile / kPartThreads, kPartRows = 3072;

__global__ __launch_bounds__(kPartThreads) void part_count_kernel(const int32_t *__restrict__ ci, long long nz0, long long n, uint32_t width, int P, uint32_t ntiles,
                                                                   uint32_t *__restrict__ cnt)
{
    __shared__ uint32_t h[kMaxSplitPanels];
    const uint32_t tile = blockIdx.x;
    if (threadIdx.x < kMaxSplitPanels) h[threadIdx.x] = 0;
    __syncthreads();
    const long long t0 = (long long)tile * kPartTile;
#pragma unroll
    for (int q = 0; q < kPartPer; q++) {
        const long long t = t0 + q * kPartThreads + threadIdx.x;
        if (t < n) atomicAdd(&h[(uint32_t)ci[nz0 + t] / width], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < P) cnt[(size_t)threadIdx.x * ntiles + tile] = h[threadIdx.x];
}

template <typename T>
__global__ __launch_bounds__(kPartThreads) void part_scatter_kernel(const long long *__restrict__ rp, long long nrows, const int32_t *__restrict__ ci, const T *__restrict__ va,
                                                                     long long nz0, long long n, uint32_t width, int P, uint32_t ntiles, const uint32_t *__restrict__ base,
                                                                     int32_t *__restrict__ ci_s, T *__restrict__ va_s, uint32_t *__restrict__ row_s)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t hist[];                  // [P][kPartThreads], then pstart[P] (u32), then the staged tile
    uint32_t *pstart = reinterpret_cast<uint32_t *>(hist + (size_t)P * kPartThreads);
    __shared__ uint32_t wsum[kPartThreads / 64];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const long long t0 = (long long)tile * kPartTile + (long long)tid * kPartPer;
    for (int p = 0; p < P; p++) hist[p * kPartThreads + tid] = 0;
    int32_t c[kPartPer];
    T       v[kPartPer];
    uint8_t pan[kPartPer];
#pragma unroll
    for (int q = 0; q < kPartPer; q++) {
        const long long t = t0 + q;
        c[q] = t < n ? ci[nz0 + t] : 0;
        v[q] = t < n ? va[nz0 + t] : T(0);
    }
#pragma unroll
    for (int q = 0; q < kPartPer; q++) {
        pan[q] = (uint8_t)((uint32_t)c[q] / width);
        if (t0 + q < n) hist[pan[q] * kPartThreads + tid] += 1;
    }
    // Rows.  The row of an element j is the last r with rp[r] <= j.  The workgroup finds the rows of the tile's first and of its last element
    // together (one half each, 128 probes per step: 4 dependent loads where a binary search per thread takes 23), brings the row pointers
    // between them into LDS (relative to the tile's first element), and every thread then looks its first row up there; its later elements
    // follow from the same copy.  A tile that spans more rows than the copy holds (long runs of empty rows) searches row_ptr itself.
    __shared__ long long s_lo[2], s_hi[2];
    __shared__ unsigned long long s_best[2];
    __shared__ int32_t   srp[kPartRows];
    const long long tile0 = (long long)tile * kPartTile, tile_n = n - tile0 < kPartTile ? n - tile0 : (long long)kPartTile;
    {
        const uint32_t  g = tid >> 7, m = tid & 127u;
        const long long j = nz0 + tile0 + (g ? tile_n - 1 : 0);
        if (m == 0) { s_lo[g] = 0; s_hi[g] = nrows; }
        __syncthreads();
        while (s_hi[0] > s_lo[0] || s_hi[1] > s_lo[1]) {                 // (uniform: both halves take part in every barrier)
            const long long lo = s_lo[g], hi = s_hi[g], step = (hi - lo) / 128 + 1, rm = lo + (long long)m * step;
            if (m == 0) s_best[g] = (unsigned long long)lo;
            __syncthreads();
            if (m > 0 && rm <= hi && rp[rm] <= j) atomicMax(&s_best[g], (unsigned long long)rm);
            __syncthreads();
            if (m == 0) { const long long b = (long long)s_best[g]; s_lo[g] = b; s_hi[g] = b + step - 1 < hi ? b + step - 1 : hi; }
            __syncthreads();
        }
    }
    const long long r_first = s_lo[0], nr = s_lo[1] - r_first + 2;      // rp[r_first .. r_last + 1]
    const bool      rows_in_lds = nr <= (long long)kPartRows;
    if (rows_in_lds)
        for (long long i = tid; i < nr; i += kPartThreads) {
            const long long d = rp[r_first + i] - (nz0 + tile0);
            srp[i] = d < -1 ? -1 : d > (1 << 30) ? (1 << 30) : (int32_t)d;
        }
    __syncthreads();
    long long r = 0;
    int32_t   li = 0;                                                    // r - r_first
    if (t0 < n) {
        if (rows_in_lds) {
            const int32_t jl = (int32_t)(tid * kPartPer);
            int32_t       lo = 0, hi = (int32_t)nr - 2;
            while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (srp[mid] <= jl) lo = mid; else hi = mid - 1; }
            li = lo; r = r_first + lo;
        } else {
            const long long j = nz0 + t0;
            long long lo = 0, hi = nrows;
            while (lo < hi) { const long long mid = (lo + hi + 1) >> 1; if (rp[mid] <= j) lo = mid; else hi = mid - 1; }
            r = lo;
        }
    }
    // ranks: hist in (panel, thread) order -- a thread sums P consecutive entries, the sums are scanned over the workgroup
    uint32_t mine = 0;
    for (int q = 0; q < P; q++) mine += hist[tid * P + q];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o); if ((int)lane >= o) incl += u; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t w = 0; w < wv; w++) run += wsum[w];
    for (int q = 0; q < P; q++) { const uint32_t k = hist[tid * P + q]; hist[tid * P + q] = (uint16_t)run; run += k; }
    __syncthreads();
    if ((int)tid < P) pstart[tid] = hist[tid * kPartThreads];          // elements of the tile in the panels in front of panel tid
    __syncthreads();
    // the elements go through LDS in their order inside the tile's share of every panel (slot = the rank the scan gave), so that the stores
    // below run along each panel's array instead of hopping between the panels element by element (soc-LiveJournal1 shape, 16 panels: 1.17 -> 0.xx ms)
    T        *sv = reinterpret_cast<T *>(reinterpret_cast<uint8_t *>(hist) + (((size_t)P * kPartThreads * sizeof(uint16_t) + (size_t)P * sizeof(uint32_t) + 7) & ~(size_t)7));
    int32_t  *sc = reinterpret_cast<int32_t *>(sv + kPartTile);
    uint32_t *sr = reinterpret_cast<uint32_t *>(sc + kPartTile);
    long long rnext = rows_in_lds ? (long long)srp[li + 1] + nz0 + tile0 : r < nrows ? rp[r + 1] : 0x7fffffffffffffffll;
#pragma unroll
    for (int q = 0; q < kPartPer; q++) {
        const long long t = t0 + q;
        if (t >= n) break;
        const long long j = nz0 + t;
        if (rows_in_lds) while (j >= rnext) { r++; li++; rnext = (long long)srp[li + 1] + nz0 + tile0; }      // (empty rows are stepped over)
        else while (j >= rnext) { r++; rnext = rp[r + 1]; }
        const uint32_t p = pan[q];
        const uint32_t slot = hist[p * kPartThreads + tid];
        hist[p * kPartThreads + tid] = (uint16_t)(slot + 1);
        sc[slot] = c[q]; sv[slot] = v[q]; sr[slot] = (uint32_t)r;
    }
    __syncthreads();
    const long long left = n - (long long)tile * kPartTile;
    const uint32_t  cnt = left < kPartTile ? (uint32_t)left : (uint32_t)kPartTile;
#pragma unroll
    for (int q = 0; q < kPartPer; q++) {
        const uint32_t i = (uint32_t)q * kPartThreads + tid;
        if (i >= cnt) break;
        int lo = 0, hi = P - 1;                                        // the panel of slot i: the last p with pstart[p] <= i
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pstart[mid] <= i) lo = mid; else hi = mid - 1; }
        const size_t pos = (size_t)base[(size_t)lo * ntiles + tile] + (i - pstart[lo]);
        ci_s[pos] = sc[i]; va_s[pos] = sv[i]; row_s[pos] = sr[i];
    }
}

// sub-row heads from the partitioned arrays themselves: a new row, or a new panel
__global__ __launch_bounds__(256) void part_head_kernel(const int32_t *__restrict__ ci_s, const uint32_t *__restrict__ row_s, long long n, uint32_t width,
                                                        uint32_t *__restrict__ head)
{
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256)
        head[t] = (t == 0 || (uint32_t)ci_s[t] / width != (uint32_t)ci_s[t - 1] / width || row_s[t] != row_s[t - 1]) ? 1u : 0u;
}

// off[p] = where panel p starts among the partitioned non-zeros (the scan's entry of tile 0), off[P] = n
__global__ void part_bounds_kernel(const uint32_t *__restrict__ base, uint32_t ntiles, long long n, int P, long long *__restrict__ off)
{
    const int p = threadIdx.x;
    if (p < P) off[p] = base[(size_t)p * ntiles];
    if (p == P) off[p] = n;
}

// what the host needs of the split, in one block: small[0] = sub-rows in front of the last element, [1] = the last element starts one,
// [2 + p] = off[p] (p = 0 .. P), [3 + P + p] = sub-rows in front of panel p's first element (one copy instead of P + 3)
__global__ void split_collect_kernel(const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ head, const long long *__restrict__ off, long long n, int P,
                                     long long *__restrict__ small)
{
    const int p = threadIdx.x;
    if (p == 0) { small[0] = sidx[n - 1]; small[1] = head[n - 1]; }
    if (p <= P) {
        small[2 + p] = off[p];
        small[3 + P + p] = off[p] < n ? (long long)sidx[off[p]] : -1;
    }
}

// ---- the panel rule's L2 model on the device (the host form: l2_miss_estimate in cvr_capi.hip) ----
constexpr uint32_t kEstBins = 4096;        // line counts 1 .. 4094 have a bin each; larger ones are summed exactly on the side

// window w = rows [r0[w], r0[w] + W): one count per 128-byte line of x
__global__ __launch_bounds__(256) void est_count_kernel(const long long *__restrict__ rp, const int32_t *__restrict__ ci, const long long *__restrict__ r0,
                                                        long long W, uint32_t per_line, long long nlines, uint32_t *__restrict__ cnt,
                                                        unsigned long long *__restrict__ refs, uint32_t nwin)
{
    // (a 1-D grid with the window in the low bits of the workgroup number: workgroups go round the XCDs, so with eight windows every
    // window's counters are touched from one XCD only)
    const uint32_t  w = blockIdx.x % nwin, bx = blockIdx.x / nwin, gx = gridDim.x / nwin;
    const long long j0 = rp[r0[w]], j1 = rp[r0[w] + W];
    uint32_t       *c = cnt + (size_t)w * (size_t)nlines;
    for (long long j = j0 + (long long)bx * 256 + threadIdx.x; j < j1; j += (long long)gx * 256) atomicAdd(&c[(uint32_t)ci[j] / per_line], 1u);
    if (bx == 0 && threadIdx.x == 0) refs[w] = (unsigned long long)(j1 - j0);
}

// The same counts without global atomics (round 5: scattered atomics run at 16-24 G/s on this chip -- 7.5 M of them were 0.38-0.47 ms of the
// soc-LiveJournal1 shape's panel rule): workgroup (q, w) reads ALL of window w's column indices and counts the lines of its range q in LDS, then
// stores its range of the counters.  For up to 32 ranges (what the windows' indices are re-read); wider matrices keep the atomic kernel.
constexpr uint32_t kEstRangeLines = 16384;          // 64 KiB of LDS counters: two workgroups per CU
__global__ __launch_bounds__(1024) void est_count_lds_kernel(const long long *__restrict__ rp, const int32_t *__restrict__ ci, const long long *__restrict__ r0,
                                                             long long W, uint32_t per_line, long long nlines, uint32_t *__restrict__ cnt,
                                                             unsigned long long *__restrict__ refs)
{
    __shared__ uint32_t c[kEstRangeLines];
    const uint32_t  q = blockIdx.x, w = blockIdx.y, lo = q * kEstRangeLines;
    const long long j0 = rp[r0[w]], j1 = rp[r0[w] + W];
    for (uint32_t i = threadIdx.x; i < kEstRangeLines; i += 1024u) c[i] = 0;
    __syncthreads();
    // 16-byte loads, four in flight per thread (one 4-byte load at a time this pass waits for memory: 0.3 ms; eight of them: 0.19): the indices in front of
    // the first 16-byte boundary and behind the last one go one by one
    auto count = [&](int32_t v) { const uint32_t l = (uint32_t)v / per_line - lo; if (l < kEstRangeLines) atomicAdd(&c[l], 1u); };      // (below the range: wraps to a large number)
    const long long ja = std::min<long long>(j1, j0 + (long long)((4u - (uint32_t)((reinterpret_cast<uintptr_t>(ci + j0) >> 2) & 3u)) & 3u)), nq = (j1 - ja) / 4, jb = ja + 4 * nq;
    for (long long j = j0 + threadIdx.x; j < ja; j += 1024) count(ci[j]);
    for (long long j = jb + threadIdx.x; j < j1; j += 1024) count(ci[j]);
    const int4 *q4 = reinterpret_cast<const int4 *>(ci + ja);
    for (long long i = threadIdx.x; i < nq; i += 1024 * 4) {
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const long long ii = i + (long long)u * 1024; v[u] = ii < nq ? q4[ii] : int4{-1, -1, -1, -1}; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (v[u].x >= 0) { count(v[u].x); count(v[u].y); count(v[u].z); count(v[u].w); }
    }
    __syncthreads();
    uint32_t *out = cnt + (size_t)w * (size_t)nlines + lo;
    for (uint32_t i = threadIdx.x; i < kEstRangeLines && (long long)lo + i < nlines; i += 1024u) out[i] = c[i];
    if (q == 0 && threadIdx.x == 0) refs[w] = (unsigned long long)(j1 - j0);
}

// per window: how many lines were touched c times (c < kEstBins - 1), and number and sum of the larger counts
__global__ __launch_bounds__(256) void est_hist_kernel(const uint32_t *__restrict__ cnt, long long nlines, uint32_t *__restrict__ hist,
                                                       unsigned long long *__restrict__ big)
{
    __shared__ uint32_t h[kEstBins];
    const uint32_t  w = blockIdx.y;
    const uint32_t *c = cnt + (size_t)w * (size_t)nlines;
    for (uint32_t i = threadIdx.x; i < kEstBins; i += 256) h[i] = 0;
    __syncthreads();
    for (long long l = (long long)blockIdx.x * 256 + threadIdx.x; l < nlines; l += (long long)gridDim.x * 256) {
        const uint32_t v = c[l];
        if (v == 0) continue;
        if (v < kEstBins - 1) atomicAdd(&h[v], 1u);
        else { atomicAdd(&big[2 * w], 1ull); atomicAdd(&big[2 * w + 1], (unsigned long long)v); }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kEstBins; i += 256) if (h[i]) atomicAdd(&hist[(size_t)w * kEstBins + i], h[i]);
}

uint32_t grid_for(long long n) { return (uint32_t)std::min<long long>(4096, std::max<long long>(1, (n + 255) / 256)); }

struct Tmp {            // frees whatever was allocated, on every path; memory the caller lent is carved first (cvr_kernels.h: Scratch)
    std::vector<void *> p;
    Scratch             lent;
    size_t              used = 0;
    template <typename U> hipError_t alloc(U **q, size_t bytes)
    {
        const size_t b = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
        if (lent.p && used + b <= lent.bytes) { *q = reinterpret_cast<U *>(static_cast<uint8_t *>(lent.p) + used); used += b; return hipSuccess; }
        hipError_t e = hipMalloc(q, b);
        if (e == hipSuccess) p.push_back(*q);
        return e;
    }
    ~Tmp() { for (void *q : p) (void)hipFree(q); }
};

}  // namespace

// Per window w (rows r0[w] .. r0[w] + W of a device-resident CSR): refs[w] = its non-zeros, hits[w] = the gathers that find their
// 128-byte line of x among the `resident` most used lines of the window, first touches excluded -- the same integers the host
// estimator computes (sum of the top counts minus one each), through a histogram of the line counts instead of a selection.
hipError_t l2_hits_device(const int64_t *rp_dev, const int32_t *ci_dev, const int64_t *r0_host, int nwin, int64_t W, int64_t ncols, bool f32, size_t resident,
                          double *refs, double *hits, hipStream_t st, Scratch lent)
{
    if (nwin < 1 || nwin > 64) return hipErrorInvalidValue;
    const uint32_t  per_line = f32 ? 32 : 16;
    const long long nlines = ncols / per_line + 1;
    Tmp             tmp;
    tmp.lent = lent;
    uint32_t           *cnt = nullptr, *hist = nullptr;
    unsigned long long *small = nullptr;     // [nwin] refs, [2 nwin] big, then the windows' first rows
    hipError_t e = tmp.alloc(&cnt, sizeof(uint32_t) * (size_t)nwin * (size_t)nlines);
    if (e == hipSuccess) e = tmp.alloc(&hist, sizeof(uint32_t) * (size_t)nwin * kEstBins);
    if (e == hipSuccess) e = tmp.alloc(&small, sizeof(unsigned long long) * 4 * (size_t)nwin);
    if (e != hipSuccess) return e;
    unsigned long long *d_refs = small, *d_big = small + nwin;
    long long          *d_r0 = reinterpret_cast<long long *>(small + 3 * nwin);
    const uint32_t nranges = (uint32_t)((nlines + kEstRangeLines - 1) / kEstRangeLines);
    const bool     in_lds = nranges <= 32 && !cvr::debug_env("est_atomics");          // (every counter is stored by its range's workgroup: no zero fill)
    e = in_lds ? hipSuccess : hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (size_t)nwin * (size_t)nlines, st);
    if (e == hipSuccess) e = hipMemsetAsync(hist, 0, sizeof(uint32_t) * (size_t)nwin * kEstBins, st);
    if (e == hipSuccess) e = hipMemsetAsync(small, 0, sizeof(unsigned long long) * 3 * (size_t)nwin, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_r0, r0_host, sizeof(long long) * (size_t)nwin, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    if (in_lds) hipLaunchKernelGGL(est_count_lds_kernel, dim3(nranges, (uint32_t)nwin), dim3(1024), 0, st, reinterpret_cast<const long long *>(rp_dev), ci_dev, d_r0, (long long)W, per_line,
                                   nlines, cnt, d_refs);
    else hipLaunchKernelGGL(est_count_kernel, dim3(512 * (uint32_t)nwin), dim3(256), 0, st, reinterpret_cast<const long long *>(rp_dev), ci_dev, d_r0, (long long)W, per_line,
                            nlines, cnt, d_refs, (uint32_t)nwin);
    hipLaunchKernelGGL(est_hist_kernel, dim3(64, (uint32_t)nwin), dim3(256), 0, st, cnt, nlines, hist, d_big);
    std::vector<uint32_t>           h((size_t)nwin * kEstBins);
    std::vector<unsigned long long> sm(3 * (size_t)nwin);
    e = hipMemcpyAsync(h.data(), hist, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(sm.data(), small, sizeof(unsigned long long) * sm.size(), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    for (int w = 0; w < nwin; w++) {
        refs[w] = (double)sm[(size_t)w];
        unsigned long long left = resident, hit = 0;
        const unsigned long long nbig = sm[(size_t)nwin + 2 * w], sbig = sm[(size_t)nwin + 2 * w + 1];
        if (nbig <= left) { hit += sbig - nbig; left -= nbig; }
        else { hit += (unsigned long long)((double)sbig / (double)nbig * (double)left) - left; left = 0; }      // (more such lines than the cache holds: their mean)
        for (uint32_t v = kEstBins - 2; v >= 1 && left > 0; v--) {
            const unsigned long long take = std::min<unsigned long long>(left, h[(size_t)w * kEstBins + v]);
            hit += take * (v - 1);
            left -= take;
        }
        hits[w] = (double)hit;
    }
    return hipSuccess;
}

// (row, panel) pairs of the windows' rows -- the partial sums column panels of `width` columns would write and the combine pass would read --
// for rows whose columns ascend (an unsorted row counts more pairs than it has: the estimate then errs against panels).  A pair begins
// where the panel changes between neighbouring non-zeros, or where a row begins inside a run of one panel.
__global__ __launch_bounds__(256) void est_pairs_kernel(const long long *__restrict__ rp, const int32_t *__restrict__ ci, const long long *__restrict__ r0, long long W,
                                                        uint32_t width, unsigned long long *__restrict__ pairs, unsigned long long *__restrict__ refs)
{
    const uint32_t  w = blockIdx.y;
    const long long ra = r0[w], j0 = rp[ra], j1 = rp[ra + W];
    if (blockIdx.x == 0 && threadIdx.x == 0) refs[w] = (unsigned long long)(j1 - j0);
    unsigned long long n = 0;
    for (long long j = j0 + 1 + (long long)blockIdx.x * 256 + threadIdx.x; j < j1; j += (long long)gridDim.x * 256) n += (uint32_t)ci[j] / width != (uint32_t)ci[j - 1] / width;
    for (long long r = ra + (long long)blockIdx.x * 256 + threadIdx.x; r < ra + W; r += (long long)gridDim.x * 256) {
        const long long s = rp[r];
        if (s < rp[r + 1] && (s == j0 || (uint32_t)ci[s] / width == (uint32_t)ci[s - 1] / width)) n++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&pairs[w], n);
}

hipError_t panel_pairs_device(const int64_t *rp_dev, const int32_t *ci_dev, const int64_t *r0_host, int nwin, int64_t W, int64_t width, double *pairs, double *refs, hipStream_t st,
                              Scratch lent)
{
    if (nwin < 1 || nwin > 64 || width < 1) return hipErrorInvalidValue;
    Tmp                 tmp;
    tmp.lent = lent;
    unsigned long long *small = nullptr;     // [nwin] pairs, [nwin] non-zeros, then the windows' first rows
    hipError_t e = tmp.alloc(&small, sizeof(unsigned long long) * 3 * (size_t)nwin);
    if (e != hipSuccess) return e;
    long long *d_r0 = reinterpret_cast<long long *>(small + 2 * nwin);
    e = hipMemsetAsync(small, 0, sizeof(unsigned long long) * 2 * (size_t)nwin, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_r0, r0_host, sizeof(long long) * (size_t)nwin, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(est_pairs_kernel, dim3(256, (uint32_t)nwin), dim3(256), 0, st, reinterpret_cast<const long long *>(rp_dev), ci_dev, d_r0, (long long)W,
                       (uint32_t)std::min<int64_t>(width, 0x7fffffff), small, small + nwin);
    std::vector<unsigned long long> sm(2 * (size_t)nwin);
    e = hipMemcpyAsync(sm.data(), small, sizeof(unsigned long long) * sm.size(), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    for (int w = 0; w < nwin; w++) { pairs[w] = (double)sm[(size_t)w]; refs[w] = (double)sm[(size_t)(nwin + w)]; }
    return hipSuccess;
}

void free_device_split(DeviceSplit &s)
{
    (void)hipFree(s.ci); (void)hipFree(s.va); (void)hipFree(s.rows); (void)hipFree(s.rp);
    s.ci = nullptr; s.va = nullptr; s.rows = nullptr; s.rp = nullptr;
}

#define SPLIT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { free_device_split(*out); return e_; } } while (0)

hipError_t split_panels_device(const int64_t *rp_dev, const int32_t *ci_dev, const void *va_dev, bool f32, int64_t nrows, int64_t nz0,
                               int64_t nz1, int64_t width, int P, DeviceSplit *out, hipStream_t st)
{
    *out = DeviceSplit();
    const long long n = nz1 - nz0;
    if (P < 1 || P > kMaxSplitPanels || width < 1 || n >= (1ll << 32) || nrows >= (1ll << 32)) return hipErrorInvalidValue;
    out->nnz = n;
    const size_t vsz = f32 ? 4 : 8;
    // the temporaries (22 bytes per non-zero + the sort's work space) in one allocation: on some boxes of the pool a single
    // hipMalloc / hipFree of such a buffer takes tens of milliseconds
    Tmp tmp;
    uint8_t  *key_in = nullptr, *key = nullptr, *arena = nullptr;
    uint32_t *idx_in = nullptr, *idx = nullptr, *row_s = nullptr, *head = nullptr, *sidx = nullptr;
    long long *off_dev = nullptr;
    int       bits = 1;
    while ((1 << bits) < P) bits++;
    size_t sort_bytes = 0, scan_bytes = 0;
    SPLIT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, key_in, key, idx_in, idx, (unsigned int)std::max<long long>(n, 1), 0, bits, st));
    SPLIT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, head, sidx, (unsigned int)std::max<long long>(n, 1), st));
    const size_t part_tiles = (size_t)((std::max<long long>(n, 1) + kPartTile - 1) / kPartTile), part_n = (size_t)P * part_tiles;
    size_t       part_scan_bytes = 0;
    SPLIT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, part_scan_bytes, head, sidx, (unsigned int)part_n, st));
    auto         up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const bool by_sort = cvr::debug_env("split_sort") != nullptr;      // (diagnostics: the sort-and-gather form)
    const bool   partition = !by_sort && width < (1ll << 31);
    if (partition) sort_bytes = 0;                                        // (keys, positions and the sort's work space -- 10 B per non-zero and more -- are the sort form's only)
    const size_t nn = (size_t)std::max<long long>(n, 4), ns = partition ? 4 : nn;
    const size_t o_key_in = 0, o_key = o_key_in + up(ns), o_idx_in = o_key + up(ns), o_idx = o_idx_in + up(4 * ns), o_row = o_idx + up(4 * ns),
                 o_head = o_row + up(4 * nn), o_sidx = o_head + up(4 * nn), o_off = o_sidx + up(4 * nn), o_part = o_off + up(sizeof(long long) * 3 * (kMaxSplitPanels + 2) + 64), o_work = o_part + up(8 * part_n),
                 total = o_work + up(std::max(std::max(sort_bytes, scan_bytes), part_scan_bytes));
    SPLIT_TRY(tmp.alloc(&arena, total));
    key_in = arena + o_key_in; key = arena + o_key;
    idx_in = reinterpret_cast<uint32_t *>(arena + o_idx_in); idx = reinterpret_cast<uint32_t *>(arena + o_idx); row_s = reinterpret_cast<uint32_t *>(arena + o_row);
    head = reinterpret_cast<uint32_t *>(arena + o_head); sidx = reinterpret_cast<uint32_t *>(arena + o_sidx); off_dev = reinterpret_cast<long long *>(arena + o_off);
    void *work = arena + o_work;
    SPLIT_TRY(hipMalloc(&out->ci, std::max<size_t>(4 * (size_t)n, 16)));
    SPLIT_TRY(hipMalloc(&out->va, std::max<size_t>(vsz * (size_t)n, 16)));
    for (int p = 0; p <= P; p++) { out->off[p] = 0; out->sub0[p] = 0; }
    if (n == 0) {
        SPLIT_TRY(hipMalloc(&out->rows, 16)); SPLIT_TRY(hipMalloc(&out->rp, 16));
        SPLIT_TRY(hipMemsetAsync(out->rp, 0, 16, st));
        return hipStreamSynchronize(st);
    }
    if (partition && n > 0) {
        const uint32_t ntiles = (uint32_t)((n + kPartTile - 1) / kPartTile);
        uint32_t      *cnt = reinterpret_cast<uint32_t *>(arena + o_part), *base = cnt + (size_t)P * ntiles;
        hipLaunchKernelGGL(part_count_kernel, dim3(ntiles), dim3(kPartThreads), 0, st, ci_dev, (long long)nz0, n, (uint32_t)width, P, ntiles, cnt);
        SPLIT_TRY(hipcub::DeviceScan::ExclusiveSum(work, part_scan_bytes, cnt, base, (unsigned int)((size_t)P * ntiles), st));
        const size_t lds = sizeof(uint16_t) * (size_t)P * kPartThreads + sizeof(uint32_t) * (size_t)P + (size_t)kPartTile * (8 + vsz) + 8;      // hist, pstart, the staged tile
        if (lds > (size_t)48 * 1024) {        // (many panels: more dynamic LDS than a kernel gets without asking)
            (void)hipFuncSetAttribute(f32 ? reinterpret_cast<const void *>(&part_scatter_kernel<float>) : reinterpret_cast<const void *>(&part_scatter_kernel<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (f32) hipLaunchKernelGGL(part_scatter_kernel<float>, dim3(ntiles), dim3(kPartThreads), lds, st, (const long long *)rp_dev, (long long)nrows, ci_dev, static_cast<const float *>(va_dev),
                                    (long long)nz0, n, (uint32_t)width, P, ntiles, base, out->ci, static_cast<float *>(out->va), row_s);
        else hipLaunchKernelGGL(part_scatter_kernel<double>, dim3(ntiles), dim3(kPartThreads), lds, st, (const long long *)rp_dev, (long long)nrows, ci_dev, static_cast<const double *>(va_dev),
                                (long long)nz0, n, (uint32_t)width, P, ntiles, base, out->ci, static_cast<double *>(out->va), row_s);
        hipLaunchKernelGGL(part_bounds_kernel, dim3(1), dim3(kMaxSplitPanels + 1), 0, st, base, ntiles, n, P, off_dev);
        hipLaunchKernelGGL(part_head_kernel, dim3(grid_for(n)), dim3(256), 0, st, out->ci, row_s, n, (uint32_t)width, head);
    } else {
    hipLaunchKernelGGL(split_key_kernel, dim3(grid_for(n)), dim3(256), 0, st, ci_dev, (long long)nz0, n, (long long)width, key_in, idx_in);
    SPLIT_TRY(hipcub::DeviceRadixSort::SortPairs(work, sort_bytes, key_in, key, idx_in, idx, (unsigned int)n, 0, bits, st));   // stable
    hipLaunchKernelGGL(split_bounds_kernel, dim3(1), dim3(kMaxSplitPanels + 1), 0, st, key, n, P, off_dev);
    if (f32) hipLaunchKernelGGL(split_gather_kernel<float>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const long long *)rp_dev, (long long)nrows, ci_dev, static_cast<const float *>(va_dev), (long long)nz0, n, idx, out->ci, static_cast<float *>(out->va), row_s);
    else hipLaunchKernelGGL(split_gather_kernel<double>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const long long *)rp_dev, (long long)nrows, ci_dev, static_cast<const double *>(va_dev), (long long)nz0, n, idx, out->ci, static_cast<double *>(out->va), row_s);
    hipLaunchKernelGGL(split_head_kernel, dim3(grid_for(n)), dim3(256), 0, st, key, row_s, n, head);
    }
    SPLIT_TRY(hipcub::DeviceScan::ExclusiveSum(work, scan_bytes, head, sidx, (unsigned int)n, st));
    long long small_host[2 * (kMaxSplitPanels + 2) + 2];
    long long *small_dev = off_dev + (kMaxSplitPanels + 2);
    hipLaunchKernelGGL(split_collect_kernel, dim3(1), dim3(kMaxSplitPanels + 1), 0, st, sidx, head, off_dev, n, P, small_dev);
    SPLIT_TRY(hipMemcpyAsync(small_host, small_dev, sizeof(long long) * (size_t)(2 * (P + 1) + 2), hipMemcpyDeviceToHost, st));
    SPLIT_TRY(hipStreamSynchronize(st));
    const long long nsub = small_host[0] + small_host[1];
    out->nsub = nsub;
    SPLIT_TRY(hipMalloc(&out->rows, std::max<size_t>(4 * (size_t)nsub, 16)));
    SPLIT_TRY(hipMalloc(&out->rp, sizeof(long long) * ((size_t)nsub + 1)));
    hipLaunchKernelGGL(split_emit_kernel, dim3(grid_for(n)), dim3(256), 0, st, head, sidx, row_s, n, out->rows, (long long *)out->rp, nsub);
    // first sub-row of every panel = the number of sub-row heads before its first element
    for (int p = 0; p <= P; p++) out->off[p] = small_host[2 + p];
    for (int p = 0; p < P; p++) out->sub0[p] = out->off[p] < n ? (int64_t)small_host[3 + P + p] : nsub;
    out->sub0[P] = nsub;
    SPLIT_TRY(hipStreamSynchronize(st));          // (the temporaries go with this call)
    SPLIT_TRY(hipGetLastError());
    return hipSuccess;
}

}  // namespace cvr

// tools/ubench/tile_spmv.hip -- prototype of the NEAR part of the near / far form (Next 3): the non-zeros of a band around the
// diagonal, shaped like the near half of the soc-LiveJournal1 stand-in (offsets Laplace-distributed with scale 50 000), in blocks of R rows x tiles
// of W columns.  A workgroup owns a block: its rows' sums live in LDS, the tiles of x are staged into LDS one after the other (coalesced loads, two
// buffers), and every gather is a ds_read.  Only (block, tile) pairs with at least twice as many entries as staging the tile costs L2 requests are
// kept; the rest would stay with the far part (column panels).  Entries of a pair are sorted by row and cut into slices of 64 (one per wavefront
// instruction); a row's entries never cross a slice, so a row gets exactly one addition per tile, in tile order: bitwise reproducible.
// The question it answers: how long does the near part take when none of its gathers goes to an L2 (138 us as one plain image of the library).
// hipcc --offload-arch=gfx950 -O3 tile_spmv.hip -o tile_spmv ; ./tile_spmv [rows] [R] [W]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr int kThreads = 1024, kWaves = kThreads / 64, kAhead = 8;

struct Tile { uint32_t col0, slice0, slice1, pad; };      // first column, slices [slice0, slice1) of 64 entries each

// meta: bits 0..11 column inside the tile, bits 12..25 row inside the block (R = the dump row of pad entries); code: index into the dictionary.
// The slices of a block's tiles are consecutive; wavefront w takes slices S0 + w, S0 + w + 16, ... whatever tile they belong to, with the loads of the
// next kAhead of them in flight, and walks through the tiles (one barrier each: the next tile of x is written into the other buffer in front of it).
template <int R, int W, bool kScan>
__global__ __launch_bounds__(kThreads) void tile_spmv_kernel(const uint32_t *__restrict__ meta, const uint8_t *__restrict__ code, const Tile *__restrict__ tiles,
                                                            const uint32_t *__restrict__ tile_ptr, const double *__restrict__ x, const double *__restrict__ dict_g,
                                                            double *__restrict__ y, uint32_t nrows, uint32_t ncols, unsigned long long *__restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    double *const acc = reinterpret_cast<double *>(smem);                  // R + 8
    double *const dict = acc + R + 8;                                      // 16
    double *const tile = dict + 16;                                        // 2 x W
    constexpr int  kPre = W / (kThreads * 2) > 0 ? W / (kThreads * 2) : 1;
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (uint32_t)R + 8; i += kThreads) acc[i] = 0.0;
    if (threadIdx.x < 16) dict[threadIdx.x] = dict_g[threadIdx.x];
    const uint32_t t0 = tile_ptr[b], t1 = tile_ptr[b + 1];
    __shared__ Tile trec[264];                 // the block's tile records (a scalar load from memory per tile and use costs ~700 clocks each)
    for (uint32_t i = threadIdx.x; i < t1 - t0 && i < 264u; i += kThreads) trec[i] = tiles[t0 + i];
    __syncthreads();
    if (t0 == t1) {
        for (uint32_t i = threadIdx.x; i < (uint32_t)R; i += kThreads) if ((size_t)b * R + i < nrows) y[(size_t)b * R + i] = 0.0;
        return;
    }
    constexpr int kDist = 3;                 // tiles of x in flight (registers) beside the two in LDS
    double2 pre[kDist][kPre];
    auto fetch = [&](double2 *dst, uint32_t t) {            // W values of x from tile t's first column on (past the end: zeros), into registers
        const uint32_t c0 = t < t1 ? trec[t - t0].col0 : 0u;
#pragma unroll
        for (int u = 0; u < kPre; u++) {
            const uint32_t i = (uint32_t)u * kThreads * 2 + threadIdx.x * 2, c = c0 + i;
            dst[u] = double2{0.0, 0.0};
            if (t < t1 && i < (uint32_t)W) {
                if (c + 1 < ncols) dst[u] = *reinterpret_cast<const double2 *>(x + c);
                else if (c < ncols) dst[u].x = x[c];
            }
        }
    };
    auto put = [&](const double2 *src, double *dst) {
#pragma unroll
        for (int u = 0; u < kPre; u++) {
            const uint32_t i = (uint32_t)u * kThreads * 2 + threadIdx.x * 2;
            if (i < (uint32_t)W) *reinterpret_cast<double2 *>(dst + i) = src[u];
        }
    };
    fetch(pre[0], t0); put(pre[0], tile);
    __syncthreads();
    uint32_t      t = t0, tend = trec[0].slice1;
    const uint32_t S0 = trec[0].slice0, S1 = trec[t1 - 1 - t0].slice1;
    const double *cur = tile;
#pragma unroll
    for (int d = 0; d < kDist; d++) fetch(pre[d], t0 + 1 + (uint32_t)d);
    unsigned long long c_bar = 0, c_load = 0, c_slice = 0;
    const unsigned long long c_start = clock64();
    auto advance = [&] {                      // on to the next tile: its x goes into the other buffer, everyone meets, one more tile is asked for
        const unsigned long long ca = clock64();
        put(pre[0], tile + (size_t)((t - t0 + 1) & 1u) * W);
        const unsigned long long cb = clock64();
        __syncthreads();
        c_load += cb - ca; c_bar += clock64() - cb;
        t++;
        cur = tile + (size_t)((t - t0) & 1u) * W;
        tend = t < t1 ? trec[t - t0].slice1 : 0xffffffffu;
#pragma unroll
        for (int d = 0; d + 1 < kDist; d++)
#pragma unroll
            for (int u = 0; u < kPre; u++) pre[d][u] = pre[d + 1][u];
        fetch(pre[kDist - 1], t + (uint32_t)kDist);
    };
    uint32_t cm[kAhead], cc[kAhead], nm[kAhead], nc[kAhead];
    auto load = [&](uint32_t *m, uint32_t *c, uint32_t k) {
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const uint32_t s = S0 + wv + (uint32_t)kWaves * (k + (uint32_t)u);
            m[u] = (uint32_t)R << 12; c[u] = 13u;
            if (s < S1) { m[u] = meta[(size_t)s * 64 + lane]; c[u] = code[(size_t)s * 64 + lane]; }
        }
    };
    load(cm, cc, 0);
    for (uint32_t k = 0; S0 + wv + (uint32_t)kWaves * k < S1; k += kAhead) {
        load(nm, nc, k + kAhead);
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const uint32_t s = S0 + wv + (uint32_t)kWaves * (k + (uint32_t)u);
            if (s < S1) {
                while (s >= tend) advance();
                const unsigned long long cs = clock64();
                const uint32_t lr = cm[u] >> 12;
                double         v = dict[cc[u]] * cur[cm[u] & 4095u];
                if (kScan) {
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {               // segmented inclusive scan over the rows of the slice (sorted)
                        const double   up = __shfl_up(v, o);
                        const uint32_t ur = __shfl_up(lr, o);
                        if (lane >= (uint32_t)o && ur == lr) v += up;
                    }
                    const uint32_t dn = __shfl_down(lr, 1);
                    if (lane == 63u || dn != lr) acc[lr] += v;        // the last entry of a row: its sum for this tile (a row's entries of a tile are in one slice)
                } else {
                    unsafeAtomicAdd(&acc[lr], v);                     // (timing only: the order of a row's additions is not fixed -- what a lane-owns-row layout would cost)
                }
                c_slice += clock64() - cs;
            }
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) { cm[u] = nm[u]; cc[u] = nc[u]; }
    }
    while (t + 1 < t1) advance();             // (every wavefront meets the others once per tile)
    __syncthreads();
    if (dbg && lane == 0 && wv == 3) { atomicAdd(dbg, clock64() - c_start); atomicAdd(dbg + 1, c_bar); atomicAdd(dbg + 2, c_load); atomicAdd(dbg + 3, c_slice); atomicAdd(dbg + 4, (unsigned long long)(t1 - t0)); }
    for (uint32_t i = threadIdx.x; i < (uint32_t)R; i += kThreads) if ((size_t)b * R + i < nrows) y[(size_t)b * R + i] = acc[i];
}

static uint64_t g_s = 88172645463325252ull;
static inline uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }

template <int R, int W, bool kScan>
int run(uint32_t n)
{
    static_assert(W % 2048 == 0 && W <= 4096, "tile staging; 12 bits of column");
    const uint32_t nblocks = (n + R - 1) / R, ntile_all = (n + W - 1) / W;
    // entries: per row 2..10 (6 on average), column = row + Laplace(50 000), as the near half of the LiveJournal stand-in
    std::vector<uint32_t> erow, ecol;
    std::vector<uint8_t>  ecode;
    erow.reserve((size_t)n * 6); ecol.reserve((size_t)n * 6); ecode.reserve((size_t)n * 6);
    for (uint32_t r = 0; r < n; r++) {
        const int k = 2 + (int)(rnd() % 9);
        for (int i = 0; i < k; i++) {
            const double u = ((rnd() >> 11) + 1) * (1.0 / 9007199254740993.0);
            int64_t      off = (int64_t)std::llround(-50000.0 * std::log(u));
            if (rnd() & 1) off = -off;
            int64_t c = (int64_t)r + off;
            if (c < 0) c = 0;
            if (c >= (int64_t)n) c = n - 1;
            erow.push_back(r); ecol.push_back((uint32_t)c); ecode.push_back((uint8_t)(rnd() % 13));
        }
    }
    const size_t ne = erow.size();
    // pairs (block, tile): counts, the good ones
    const uint32_t span = 256;              // tiles looked at per block, centred on its diagonal
    std::vector<uint32_t> cnt((size_t)nblocks * span, 0);
    auto slot = [&](uint32_t r, uint32_t c, uint32_t *out) {
        const uint32_t b = r / R;
        const int64_t  tc = (int64_t)c / W, tb = ((int64_t)b * R + R / 2) / W - span / 2, d = tc - tb;
        if (d < 0 || d >= span) return false;
        *out = b * span + (uint32_t)d;
        return true;
    };
    for (size_t e = 0; e < ne; e++) { uint32_t s; if (slot(erow[e], ecol[e], &s)) cnt[s]++; }
    const uint32_t need = 2 * (W * 8 / 128);
    size_t                kept = 0, ntiles = 0;
    for (size_t s = 0; s < cnt.size(); s++) if (cnt[s] >= need) { kept += cnt[s]; ntiles++; }
    // bucket the kept entries by pair (rows ascending inside: generated row by row)
    std::vector<uint32_t> start(cnt.size() + 1, 0);
    for (size_t s = 0; s < cnt.size(); s++) start[s + 1] = start[s] + (cnt[s] >= need ? cnt[s] : 0);
    std::vector<uint32_t> fill(start.begin(), start.end() - 1), order(kept);
    for (size_t e = 0; e < ne; e++) { uint32_t s; if (slot(erow[e], ecol[e], &s) && cnt[s] >= need) order[fill[s]++] = (uint32_t)e; }
    // slices of 64: a row's entries of a pair stay in one slice (pad entries go to the dump row R with column 0, code 13 = 0.0)
    std::vector<uint32_t> meta;
    std::vector<uint8_t>  code;
    std::vector<Tile>     tiles;
    std::vector<uint32_t> tile_ptr(nblocks + 1, 0);
    meta.reserve(kept + kept / 8); code.reserve(kept + kept / 8);
    for (uint32_t b = 0; b < nblocks; b++) {
        tile_ptr[b] = (uint32_t)tiles.size();
        for (uint32_t d = 0; d < span; d++) {
            const size_t s = (size_t)b * span + d;
            if (cnt[s] < need) continue;
            const int64_t tb = ((int64_t)b * R + R / 2) / W - span / 2;
            const uint32_t col0 = (uint32_t)((tb + d) * W);
            Tile tl{col0, (uint32_t)(meta.size() / 64), 0, 0};
            size_t i = start[s];
            while (i < start[s + 1]) {
                size_t j = i;
                while (j < start[s + 1] && erow[order[j]] == erow[order[i]]) j++;
                const size_t len = j - i, used = meta.size() % 64;
                if (len > 64) { fprintf(stderr, "row segment longer than a slice\n"); return 2; }
                if (used + len > 64) for (size_t p = used; p < 64; p++) { meta.push_back((uint32_t)R << 12); code.push_back(13); }
                for (size_t q = i; q < j; q++) {
                    const uint32_t e = order[q];
                    meta.push_back(((erow[e] - b * R) << 12) | (ecol[e] - col0));
                    code.push_back(ecode[e]);
                }
                i = j;
            }
            while (meta.size() % 64) { meta.push_back((uint32_t)R << 12); code.push_back(13); }
            tl.slice1 = (uint32_t)(meta.size() / 64);
            tiles.push_back(tl);
        }
    }
    tile_ptr[nblocks] = (uint32_t)tiles.size();
    (void)ntile_all;
    double dict[16] = {0};
    for (int i = 0; i < 13; i++) dict[i] = 1.0 + i;
    std::vector<double> x(n), yref(n, 0.0);
    for (uint32_t i = 0; i < n; i++) x[i] = (double)(rnd() % 2001) / 1000.0 - 1.0;
    for (size_t q = 0; q < kept; q++) { const uint32_t e = order[q]; yref[erow[e]] += dict[ecode[e]] * x[ecol[e]]; }

    uint32_t *d_meta, *d_tp; uint8_t *d_code; Tile *d_tiles; double *d_x, *d_y, *d_dict;
    CHECK(hipMalloc(&d_meta, meta.size() * 4)); CHECK(hipMalloc(&d_code, code.size())); CHECK(hipMalloc(&d_tiles, tiles.size() * sizeof(Tile)));
    CHECK(hipMalloc(&d_tp, tile_ptr.size() * 4)); CHECK(hipMalloc(&d_x, (size_t)n * 8 + 16)); CHECK(hipMalloc(&d_y, (size_t)n * 8)); CHECK(hipMalloc(&d_dict, 128));
    CHECK(hipMemcpy(d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_code, code.data(), code.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_tp, tile_ptr.data(), tile_ptr.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_x, x.data(), (size_t)n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_dict, dict, 128, hipMemcpyHostToDevice));
    unsigned long long *d_dbg; CHECK(hipMalloc(&d_dbg, 64)); CHECK(hipMemset(d_dbg, 0, 64));
    const size_t lds = ((size_t)R + 8 + 16 + 2 * (size_t)W) * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_spmv_kernel<R, W, kScan>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&] { hipLaunchKernelGGL((tile_spmv_kernel<R, W, kScan>), dim3(nblocks), dim3(kThreads), lds, 0, d_meta, d_code, d_tiles, d_tp, d_x, d_dict, d_y, n, n, d_dbg); };
    for (int w = 0; w < 5; w++) launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 50;
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < iters; w++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> y(n), y2(n);
    CHECK(hipMemcpy(y.data(), d_y, (size_t)n * 8, hipMemcpyDeviceToHost));
    launch(); CHECK(hipMemcpy(y2.data(), d_y, (size_t)n * 8, hipMemcpyDeviceToHost));
    double worst = 0; size_t differ = 0;
    for (uint32_t i = 0; i < n; i++) { worst = std::max(worst, std::fabs(y[i] - yref[i]) / (1.0 + std::fabs(yref[i]))); differ += y[i] != y2[i]; }
    unsigned long long dbg[5]; CHECK(hipMemcpy(dbg, d_dbg, 40, hipMemcpyDeviceToHost));
    printf("  clocks per block (wavefront 3): total %.0f, at barriers %.0f, writing the x tile %.0f, in slices %.0f; tiles per block %.1f\n", dbg[0] / (56.0 * nblocks), dbg[1] / (56.0 * nblocks), dbg[2] / (56.0 * nblocks), dbg[3] / (56.0 * nblocks), dbg[4] / (56.0 * nblocks));
    const double us = ms * 1e3 / iters, slots = (double)meta.size();
    printf("%s R %d W %d (LDS %zu KiB): rows %u, entries %zu, kept in %zu (block, tile) pairs %zu (%.1f %%; %.1f tiles per block), slots %.0f (+%.1f %% pad), "
           "staging %.1f M requests | %.1f us per SpMV = %.1f G entries/s, %.0f GB/s of image | worst rel err %.2e, rerun differs in %zu rows\n",
           kScan ? "scan" : "atomic", R, W, lds >> 10, n, ne, ntiles, kept, 100.0 * kept / ne, (double)ntiles / nblocks, slots, 100.0 * (slots - kept) / kept, ntiles * (W * 8 / 128) / 1e6, us,
           kept / us / 1e3, slots * 5 / us / 1e3, worst, differ);
    for (void *p : {(void *)d_meta, (void *)d_code, (void *)d_tiles, (void *)d_tp, (void *)d_x, (void *)d_y, (void *)d_dict}) (void)hipFree(p);
    return worst < 1e-12 && (differ == 0 || !kScan) ? 0 : 1;
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 4847571u;
    int rc = 0;
    rc |= run<8192, 4096, true>(n);
    rc |= run<8192, 4096, false>(n);
    rc |= run<4096, 4096, false>(n);
    rc |= run<12288, 2048, false>(n);
    return rc;
}

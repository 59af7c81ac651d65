// cvr_spmv.hip -- y = A x over the CVR64 image, one wavefront per chunk (gfx950, wave64).  Memory-bound: by HBM on
// matrices whose columns share lines (banded), by the CU's L1 miss path (128-byte fills) on scattered columns.
//
// The reference's spmv_compute_kernel (/root/reference/spmv.cpp:1016-1667) re-derived for 64 lanes:
//   * one loop instead of the five hand-split phases A-E (spmv.cpp:1167-1629): each step is
//     acc += val * x[col] (spmv.cpp:1226-1233).  The matrix stream arrives as 16-B-per-lane coalesced buffer loads
//     (1 KiB per wave instruction; the reference: one 64-B load per 8 lanes) two groups of 4 steps ahead; the x gather
//     (_mm512_i32logather_pd, spmv.cpp:1227) is one buffer_load_dwordx2 per lane, issued one group ahead of its use in
//     place of the software prefetch of spmv.cpp:1183-1190; matrices with <= 256 distinct values stream one code byte
//     per slot and look the value up in LDS;
//   * the scalar record-servicing `while` (spmv.cpp:1197-1224) is replaced by bit 31 of the column word: the lanes
//     whose segment ends at this step form a __ballot mask; each hands over its row sum (spmv.cpp:1204-1205
//     get_simd/set_simd_zero) and takes segment fed + rank-in-mask, the same hand-out order the converter used, so no
//     write-back ids are read from memory.  The row sums are staged in LDS by segment ordinal and leave as coalesced
//     stores at the end of the chunk (its rows are consecutive);
//   * the steal part (spmv.cpp:1579-1629) runs the SAME loop; only the write-back differs: a lane whose own row ends
//     after the last segment was handed out parks the sum in its LDS slot (t_rets, spmv.cpp:1607-1616), lanes that
//     stole add their partial sums to the victim's slot at the end (tail records, spmv.cpp:1633-1638: ds_add here);
//   * no `#pragma omp atomic` on y (spmv.cpp:1280-1282, 1640-1649) and no zeroing of y (spmv.cpp:1026-1031): chunks
//     end at row boundaries; the few rows cut over chunks go to carry slots behind y and are summed in chunk order by
//     fixup_kernel (bitwise reproducible); column panels add their partial sums in panel order (combine_kernel).
#include "cvr_kernels.h"

#include <cstdio>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace cvr {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double   f64x2 __attribute__((ext_vector_type(2)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Cache-policy bits of a gfx950 buffer load (the `aux` immediate): sc0 = 1, nt = 2, sc1 = 16.
constexpr int kPolDefault = 0;

// Both the matrix stream and x are read through buffer descriptors: 32-bit offsets, the cache policy is an
// immediate, and a load past num_records returns 0 without touching memory -- which lets the software
// pipeline run its loads unconditionally past the end of a chunk (counted vmcnt waits stay exact).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

template <typename T, bool DICT> struct Group;
template <> struct Group<double, false> { u32x4 c; f64x2 lo, hi; };
template <> struct Group<float, false>  { u32x4 c; f32x4 v; };
template <typename T> struct Group<T, true> { u32x4 c; uint32_t codes; };   // four dictionary codes, one byte per step

// C16 (narrow chunks): the group's column part is [64 lanes][4 x u16 offset from the chunk's smallest column]; the loaded
// words are widened to the usual column words (end flag in bit 31, the pad column for 0x7fff) by widen_cols once the
// data has arrived, so the rest of the kernel does not know the difference.
template <typename T, int POL, bool DICT, bool C16 = false>
__device__ __forceinline__ Group<T, DICT> load_group(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    Group<T, DICT> g;
    if constexpr (C16) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff >> 1, soff, POL));
        g.c = u32x4{h.x, h.y, 0u, 0u};
        if constexpr (sizeof(T) == 8) {
            g.lo = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + kCols16Bytes, soff, POL));
            g.hi = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + kCols16Bytes + kLanes * 16, soff, POL));
        } else {
            g.v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + kCols16Bytes, soff, POL));
        }
        return g;
    }
    g.c = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, POL));
    if constexpr (DICT) {
        g.codes = __builtin_amdgcn_raw_buffer_load_b32(r, (voff >> 2) + kColsBytes, soff, POL);
    } else if constexpr (sizeof(T) == 8) {
        g.lo = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + kColsBytes, soff, POL));
        g.hi = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + kColsBytes + kLanes * 16, soff, POL));
    } else {
        g.v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + kColsBytes, soff, POL));
    }
    return g;
}

__device__ __forceinline__ uint32_t widen_col(uint32_t h, uint32_t base, uint32_t pad_col)
{
    const uint32_t off = h & kC16Pad;
    return (off == kC16Pad ? pad_col : base + off) | ((h & 0x8000u) << 16);
}
// the two loaded words of a narrow chunk's group (four 16-bit offsets) -> four column words
__device__ __forceinline__ u32x4 widen_cols(const u32x4 c, uint32_t base, uint32_t pad_col)
{
    return u32x4{widen_col(c.x & 0xffffu, base, pad_col), widen_col(c.x >> 16, base, pad_col), widen_col(c.y & 0xffffu, base, pad_col), widen_col(c.y >> 16, base, pad_col)};
}

template <typename T> struct X4 { T v[4]; T w[4]; };   // v: through the buffer descriptor, w: from the LDS window

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float  fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T, int POL>
__device__ __forceinline__ T load_x(__amdgpu_buffer_rsrc_t rx, uint32_t col)
{
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, col * 8u, 0, POL));
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, col * 4u, 0, POL));
}

// The x values of four steps.  Without a window every lane reads x[col] through the buffer descriptor.  With a
// window [wbase, wbase + wn) staged in LDS, lanes whose column falls inside read LDS and send the global load out
// of range (returns 0, no memory traffic); the other lanes read the zero slot win[wn].  The two halves are OR-ed
// where the value is used (x_of), so that both loads stay in flight until then.
// WIN: 0 = no LDS table, 1 = a window of x, 2 = a hub table (with or without a window behind it).
template <typename T, int POL, int WIN>
__device__ __forceinline__ X4<T> gather(__amdgpu_buffer_rsrc_t rx, const T *win, const u32x4 c, const uint32_t mask,
                                        const uint32_t wbase, const uint32_t wn, const uint32_t hub_n, const uint32_t zero_at)
{
    X4<T>          r;
    const uint32_t col[4] = {c.x & mask, c.y & mask, c.z & mask, c.w & mask};
    if constexpr (WIN == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) r.v[j] = load_x<T, POL>(rx, col[j]);
    } else if constexpr (WIN == 1) {
        // LDS: [window of x (wn) | zeros]: five vector instructions per step less than the hub form (the SpMV loop of the resident
        // layout is bound by instruction issue: two wavefronts per SIMD, ~70 vector instructions per step)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t rel = col[j] - wbase;
            r.v[j] = load_x<T, POL>(rx, rel < wn ? 0x3fffffffu : col[j]);   // 0x3fffffff * sizeof(T) is past num_records
            r.w[j] = win[rel < wn ? rel : zero_at];        // (zero_at = wn, except while the window is still on its way: then wn here is 0)
        }
    } else {
        // LDS: [hub table (hub_n) | window of x (wn) | zeros].  A slot of a hub column holds kHubBit and the table index.
        const uint32_t raw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool     hub = (raw[j] & kHubBit) != 0 && hub_n != 0;
            const uint32_t cj = hub_n ? col[j] & (kHubBit - 1u) : col[j];
            const uint32_t rel = cj - wbase;
            const bool     in = rel < wn;
            r.v[j] = load_x<T, POL>(rx, hub || in ? 0x3fffffffu : cj);   // 0x3fffffff * sizeof(T) is past num_records
            r.w[j] = win[hub ? cj : hub_n + (in ? rel : wn)];
        }
    }
    return r;
}

template <typename T, int WIN>
__device__ __forceinline__ T x_of(const X4<T> &x, int j)
{
    if constexpr (WIN == 0) return x.v[j];
    else if constexpr (sizeof(T) == 8) return __builtin_bit_cast(double, __builtin_bit_cast(uint64_t, x.v[j]) | __builtin_bit_cast(uint64_t, x.w[j]));
    else return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x.v[j]) | __builtin_bit_cast(uint32_t, x.w[j]));
}

template <typename T, bool DICT>
__device__ __forceinline__ T val_of(const Group<T, DICT> &g, int j, const T *dict)
{
    if constexpr (DICT) return dict[(g.codes >> (8 * j)) & 0xffu];            // LDS lookup
    else if constexpr (sizeof(T) == 8) return j == 0 ? g.lo.x : j == 1 ? g.lo.y : j == 2 ? g.hi.x : g.hi.y;
    else return j == 0 ? g.v.x : j == 1 ? g.v.y : j == 2 ? g.v.z : g.v.w;
}

__device__ __forceinline__ uint32_t col_of(const u32x4 c, int j) { return j == 0 ? c.x : j == 1 ? c.y : j == 2 ? c.z : c.w; }

__device__ __forceinline__ uint32_t remap_block(uint32_t b, uint32_t nblocks_per_xcd, int swz)
{
    // blocks are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, "Workgroup dispatch"): give each
    // XCD one contiguous range of chunks so that neighbouring rows' x lines meet in one L2.  Speed only.
    if (swz == 0) return b;
    const uint32_t xcd = b & 7u, j = b >> 3;
    if (swz == 1) {
        // (here the argument is the TOTAL number of blocks n = 8 q + r: the first r XCDs take q + 1 consecutive blocks, the others q,
        // so that no XCD's L2 serves more workgroups than another's by more than one -- the resident layout is bound by the L2s)
        const uint32_t n = nblocks_per_xcd, q = n >> 3, r = n & 7u;
        const uint32_t cnt = q + (xcd < r ? 1u : 0u), base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
        return j < cnt ? base + j : 0x00ffffffu;
    }
    if (swz >= 3) {
        // swz = 3, 4, 5, 6: an XCD takes RUNS of 2, 4, 8, 16 consecutive blocks, the runs dealt round-robin over the XCDs -- neighbouring rows still
        // meet in one L2, and stretches of the matrix that gather more than others are spread over all eight (here the argument is the total n)
        const uint32_t lg = (uint32_t)swz - 2u, blk = ((((j >> lg) << 3) + xcd) << lg) + (j & ((1u << lg) - 1u));
        return blk < nblocks_per_xcd ? blk : 0x00ffffffu;
    }
    // swz = 2 (experiment): within the XCD, blocks j, j+32, j+64, .. (one CU under round-robin placement) take
    // consecutive chunks, so that the waves resident on a CU work on neighbouring rows and share x lines in its L1
    const uint32_t per_cu = (nblocks_per_xcd + 31) / 32;
    const uint32_t t = (j & 31u) * per_cu + (j >> 5);
    return t < nblocks_per_xcd ? xcd * nblocks_per_xcd + t : 0x00ffffffu;
}


// y stores stay plain: nontemporal stores of these scattered 8-byte values doubled the kernel time
// (profiles/r01_waves_per_block.log)
template <typename T> __device__ __forceinline__ void store_y(T *p, T v) { *p = v; }

// Wave-uniform and per-lane state of one chunk.
template <typename T> struct ChunkState {
    T        acc;       // running sum of the lane's current segment
    uint32_t cur;       // ordinal of that segment in the chunk
    uint32_t fed;       // segments handed out so far (wave-uniform)
    uint32_t feeding;   // 0: the lane is (or has become) a stealer
    uint32_t own;       // the lane parked a row sum in its LDS slot
    uint32_t tail;      // every segment handed out (wave-uniform): write-backs go through the slots
};

template <typename T> __device__ __forceinline__ void lds_add(T *p, T v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);      // ds_add_f64 / ds_add_f32
}
template <typename T> __device__ __forceinline__ void lds_add_wg(T *p, T v)      // the same instruction; accumulators that other wavefronts of the workgroup add into as well (gang chunks)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// four steps of all 64 lanes: FMA, then the write-back of the lanes whose segment ends at the step.  (Images with column phases
// run through spmv_seg_kernel below, which needs none of this hand-out state.)
template <typename T, int WIN, bool DICT>
__device__ __forceinline__ void sum_group(ChunkState<T> &s, const Group<T, DICT> &Q, const X4<T> &xq, T *__restrict__ yext,
                                          T *slot_lane, uint32_t row_first, uint32_t nseg, uint32_t head_dest,
                                          uint32_t last_dest, const T *dict, T *ystage, bool staged)
{
    // the four values first: with a dictionary they are LDS reads, which would otherwise be issued (and waited for) one by
    // one between the steps' LDS writes
    T av[kGroupSteps];
#pragma unroll
    for (int j = 0; j < kGroupSteps; j++) av[j] = val_of<T, DICT>(Q, j, dict);
#pragma unroll
    for (int j = 0; j < kGroupSteps; j++) {
        const uint32_t cw = col_of(Q.c, j);
        s.acc = fma_t(av[j], x_of<T, WIN>(xq, j), s.acc);
        const bool     fl = (cw & kEndBit) != 0;
        const uint64_t m = __ballot(fl);
        if (m) {
            if (!s.tail) {
                if (fl) {
                    if (staged) {
                        ystage[s.cur] = s.acc;           // written out coalesced at the end of the chunk
                    } else {
                        const uint32_t dst = s.cur == 0 ? head_dest : s.cur == nseg - 1 ? last_dest : row_first + s.cur;
                        store_y(yext + dst, s.acc);
                    }
                    s.acc = 0;
                    const uint32_t nx = s.fed + lane_rank(m);
                    if (nx < nseg) s.cur = nx; else s.feeding = 0;   // rows exhausted: turns stealer
                }
                s.fed = __builtin_amdgcn_readfirstlane(s.fed + (uint32_t)__popcll(m));
                if (s.fed >= nseg) { s.fed = nseg; s.tail = 1; }
            } else if (fl && s.feeding) {
                *slot_lane = s.acc;
                s.acc = 0;
                s.feeding = 0;
                s.own = 1;
            }
        }
    }
}

// MW: more than one wavefront per workgroup (blockDim.x / 64 consecutive chunks share the workgroup's LDS window of x and its
// dictionary copy); the single-wavefront form needs no barrier.
// multi != null (column panels, one panel per XCD at a time): workgroup b works on panel b & 7 of the eight the launch covers --
// the workgroups with equal b & 7 share an XCD under round-robin dealing, so that XCD's L2 only ever holds that panel's slice
// of x -- and takes the panel's chunk b >> 3; what differs between the panels comes from multi[b & 7].
template <typename T, int QA, int XPOL, int DEPTH, int WIN, bool DICT, bool MW, bool C16>
__global__ __launch_bounds__(MW ? kLanes * kMaxWavesPerBlock : kLanes) void spmv_kernel(
    const uint8_t *__restrict__ stream_a, const uint4 *__restrict__ desc_a, const uint8_t *__restrict__ target_a,
    const T *__restrict__ x, T *__restrict__ yext_a, int G, uint32_t nchunks_a, uint32_t nblocks_per_xcd, int swz,
    uint32_t cmask, uint32_t xbytes, const uint32_t *__restrict__ win_base, uint32_t wn, const T *__restrict__ dict_g, uint32_t ndict,
    uint32_t ystage_a, const T *__restrict__ hub_x, uint32_t hub_n, uint32_t kstride,
    const uint32_t *__restrict__ cbase, uint32_t pad_col, const PanelArgs *__restrict__ multi, uint32_t stream_mod)
{
    const uint8_t *__restrict__ stream = stream_a, *__restrict__ target = target_a;
    const uint4 *__restrict__   desc = desc_a;
    T *__restrict__             yext = yext_a;
    uint32_t                    nchunks = nchunks_a, ystage_n = ystage_a, bidx = blockIdx.x;
    if (multi) {          // (nblocks_per_xcd: here the workgroups of one round of eight panels; the rounds follow each other in the grid)
        const uint32_t  round = blockIdx.x / nblocks_per_xcd, b = blockIdx.x - round * nblocks_per_xcd;
        const PanelArgs pa = multi[round * 8u + (b & 7u)];
        stream = pa.stream; desc = pa.desc; target = pa.target; yext = static_cast<T *>(pa.yext); nchunks = pa.nchunks; ystage_n = pa.ystage;
        bidx = b >> 3;
    }
    constexpr int  GB = DICT ? kGroupBytesDict : C16 ? (sizeof(T) == 8 ? kGroupBytes64C16 : kGroupBytes32C16) : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32;
    constexpr bool kSync = WIN != 0 || (DICT && MW);      // LDS filled by other waves of the workgroup
    // LDS: [waves][64] steal slots, [waves][ystage_n] staged row sums, the value dictionary (DICT),
    // the x window and its zero slot (WIN; wn + 4 values)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // (Helper wavefronts as in spmv_ilv_kernel -- scalar prefetch of the stream -- were measured on this kernel's persistent hub-table form and
    // made it slower: R-MAT-22 fp32 286 -> 308-315 us, profiles/r05_rmat_helpers_not_adopted.log: the prefetched lines displace x in the L2s.)
    const uint32_t nw = MW ? blockDim.x >> 6 : 1u;
    T *const slots = reinterpret_cast<T *>(smem);
    T *const ystage_all = slots + nw * kLanes;
    T *const dict = ystage_all + nw * ystage_n;
    T *const win = dict + (DICT ? kDictMax : 0);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = MW ? threadIdx.x >> 6 : 0u;
    const uint32_t blk = remap_block(bidx, nblocks_per_xcd, swz);
    uint32_t       k = __builtin_amdgcn_readfirstlane(blk * nw + wv);
    if (!kSync && k >= nchunks) return;
    bool live = k < nchunks;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, xbytes);
    const uint32_t voff = lane * 16;
    T *const       ystage = ystage_all + wv * ystage_n;
    T *const       slot_lane = &slots[wv * kLanes + lane];

    // software pipeline: the x gather runs DEPTH groups ahead of the FMAs, the matrix stream QA groups ahead of the gather
    constexpr int  QN = DEPTH + QA;
    // (WIN == 3: the hub-table kernel of an image that does not stay in the caches between SpMVs loads its stream non-temporally, so that it leaves the
    // L2s before x does: R-MAT-22 fp32 293 -> 271 us, R-MAT-24 1 197 -> 1 133; the plain kernel of a banded matrix LOSES by it, 178 -> 197: profiles/r05_stream_nt_hub.log)
    constexpr int  SPOL = WIN == 3 ? 2 : kPolDefault;
    Group<T, DICT> Q[QN];
    X4<T>          xs[DEPTH];
    __amdgpu_buffer_rsrc_t rs;
    uint4          d;
    uint32_t       cb = 0;                             // C16: the chunk's smallest column
    // the first loads of chunk k: its stream (a wave past the last chunk streams nothing), its descriptor
    auto begin_chunk = [&]() {
        // (stream_mod, CVR_DEBUG=stream_mod=M: timing only, wrong sums -- every chunk streams the image of one of the first M chunks, i.e. an L2-resident
        // stream: what a perfect prefetch of the stream would give)
        rs = make_rsrc(stream + (size_t)(live ? (stream_mod ? k % stream_mod : k) : 0) * ((size_t)G * GB), live ? (uint32_t)G * GB : 0u);
#pragma unroll
        for (int i = 0; i < QN; i++) Q[i] = load_group<T, SPOL, DICT, C16>(rs, voff, (uint32_t)i * GB);
        d = live ? desc[k] : uint4{0, 0, 0, 0};
        if constexpr (C16) cb = live ? cbase[k] : 0u;
    };
    begin_chunk();

    // stage the dictionary, the hub table and this workgroup's window of x in LDS: coalesced 16-byte loads, behind the first stream loads
    uint32_t wbase = 0;
    if constexpr (DICT)
        for (uint32_t i = threadIdx.x; i < (uint32_t)kDictMax; i += blockDim.x) dict[i] = i < ndict ? dict_g[i] : T(0);
    if constexpr (WIN != 0) {
        constexpr uint32_t kPer = 16 / sizeof(T);                      // values per 16-byte load; wbase and wn are multiples of it
        wbase = wn && blk * nw < nchunks ? win_base[blk] : 0u;
        // eight 16-byte loads in flight per lane before the first LDS store (the 64-KB window takes 1.8 us either way -- all
        // 256 workgroups fetch theirs at once, ~9 TB/s --, profiles/r02_kernel_timeline.log)
        constexpr int kBatch = 8;
        const uint32_t hpad = (hub_n + 3u) & ~3u;
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(hub_x, hpad * (uint32_t)sizeof(T));
        const uint32_t tstep = blockDim.x * kPer;
        for (uint32_t i0 = threadIdx.x * kPer; i0 < hub_n; i0 += tstep * kBatch) {                 // the hub table (hub_x is padded to whole 16 bytes)
            u32x4 q[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) q[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rh, (i0 + u * tstep) * (uint32_t)sizeof(T), 0, kPolDefault));
#pragma unroll
            for (int u = 0; u < kBatch; u++) if (i0 + u * tstep < hub_n) *reinterpret_cast<u32x4 *>(win + i0 + u * tstep) = q[u];
        }
        for (uint32_t i0 = threadIdx.x * kPer; i0 < wn; i0 += tstep * kBatch) {
            u32x4 q[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const uint32_t i = i0 + u * tstep;
                q[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, i < wn ? (wbase + i) * (uint32_t)sizeof(T) : 0xfffffff0u, 0, kPolDefault));
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++) if (i0 + u * tstep < wn) *reinterpret_cast<u32x4 *>(win + hpad + i0 + u * tstep) = q[u];
        }
        if (threadIdx.x < 4) win[hpad + wn + threadIdx.x] = T(0);
    }
    if constexpr (kSync) {
        __syncthreads();
    } else if constexpr (DICT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // one wavefront per workgroup: its own LDS writes, in order
    }

    // Persistent workgroups (kstride > 0; hub table without a per-workgroup window): the workgroup keeps its LDS tables and
    // takes chunk groups blk, blk + gridDim.x, ...; no barrier past this point.  Otherwise one chunk per wavefront.
    for (;;) {
        if (!live) break;
        const uint32_t row_first = d.x, nseg = d.y, head_dest = d.z, last_dest = d.w;
        const uint32_t tg = target[(size_t)k * kLanes + lane];

        ChunkState<T> s;
        s.acc = 0;
        s.cur = lane;
        s.fed = nseg < kLanes ? nseg : kLanes;
        s.feeding = lane < s.fed;
        s.own = 0;
        s.tail = s.fed == nseg;
        // Row sums go to LDS and leave as coalesced stores at the end of the chunk (its rows are consecutive): the scattered
        // 8-byte stores they replace cost 12 % of the kernel (profiles/r01_y_staging.log).  A chunk of more than ystage_n
        // segments (very short rows) stores directly.
        const bool staged = nseg <= ystage_n;
        if constexpr (C16) {
#pragma unroll
            for (int i = 0; i < DEPTH; i++) Q[i].c = widen_cols(Q[i].c, cb, pad_col);
        }
#pragma unroll
        for (int i = 0; i < DEPTH; i++) xs[i] = gather<T, XPOL, WIN>(rx, win, Q[i].c, cmask, wbase, wn, (hub_n + 3u) & ~3u, wn);

        // every load is unconditional: past the end of the chunk the stream loads are out of range (zeros, no
        // traffic) and the gathers they feed all read x[0]
        for (int g = 0; g < G; g++) {
            const Group<T, DICT> Qn = load_group<T, SPOL, DICT, C16>(rs, voff, (uint32_t)(g + QN) * GB);
            if constexpr (C16) Q[DEPTH].c = widen_cols(Q[DEPTH].c, cb, pad_col);       // (arrived an iteration ago: the gather below needs it anyway)
            const X4<T>    xn = gather<T, XPOL, WIN>(rx, win, Q[DEPTH].c, cmask, wbase, wn, (hub_n + 3u) & ~3u, wn);
            sum_group<T, WIN, DICT>(s, Q[0], xs[0], yext, slot_lane, row_first, nseg, head_dest, last_dest, dict, ystage, staged);
#pragma unroll
            for (int i = 0; i + 1 < QN; i++) Q[i] = Q[i + 1];
            Q[QN - 1] = Qn;
#pragma unroll
            for (int i = 0; i + 1 < DEPTH; i++) xs[i] = xs[i + 1];
            xs[DEPTH - 1] = xn;
        }

        // tail records (spmv.cpp:1633-1638): stolen partial sums go to the victim's slot, owners store
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (tg != lane) __hip_atomic_fetch_add(&slots[wv * kLanes + tg], s.acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (s.own) {
            if (staged) {
                ystage[s.cur] = *slot_lane;
            } else {
                const uint32_t dst = s.cur == 0 ? head_dest : s.cur == nseg - 1 ? last_dest : row_first + s.cur;
                store_y(yext + dst, *slot_lane);
            }
        }
        if (staged) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const uint32_t nout = nseg;
            for (uint32_t i = lane; i < nout; i += kLanes) {
                const uint32_t dst = i == 0 ? head_dest : i == nout - 1 ? last_dest : row_first + i;
                store_y(yext + dst, ystage[i]);
            }
        }
        if (kstride == 0) break;
        k += kstride;
        live = k < nchunks;
        if (!live) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // this wave's stage and slots are reused by its next chunk
        begin_chunk();
    }
}

// ---- column phases: the kernel of the resident layout -------------------------------------------------------------------
// With column phases the last column word of EVERY piece of a lane stream -- a (row, phase) segment, or what a lane stole of
// one -- carries the chunk's row it belongs to (bits [col_bits, 31)), so a lane needs no state but its running sum: when a
// piece ends, the sum is added to the row's accumulator in LDS (ds_add, lanes in order, steps in order: reproducible) and
// the rows leave coalesced at the end of the chunk.  No ballot, no rank, no hand-out counter, no steal slots, no `target`
// (what spmv.cpp:1197-1224 and 1579-1651 do with records and t_rets): ~20 vector instructions per step where the general
// kernel above issues ~70, which is what bounds a loop of 12 groups per wavefront with two wavefronts per SIMD.
// TAG (wide row tags): the rows stand in 16-bit tags of their own, four per lane and group, instead of above the column index
template <typename T, bool DICT, bool TAG> struct SegGroup : Group<T, DICT> { uint32_t t01, t23; };

template <typename T, bool DICT, bool TAG>
__device__ __forceinline__ SegGroup<T, DICT, TAG> load_seg_group(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    SegGroup<T, DICT, TAG> g;
    constexpr uint32_t VB = kColsBytes + (TAG ? kTagBytes : 0);         // where the values / codes start
    g.c = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kPolDefault));
    if constexpr (TAG) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (voff >> 1) + kColsBytes, soff, kPolDefault));
        g.t01 = t.x; g.t23 = t.y;
    } else { g.t01 = 0; g.t23 = 0; }
    if constexpr (DICT) {
        g.codes = __builtin_amdgcn_raw_buffer_load_b32(r, (voff >> 2) + VB, soff, kPolDefault);
    } else if constexpr (sizeof(T) == 8) {
        g.lo = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + VB, soff, kPolDefault));
        g.hi = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + VB + kLanes * 16, soff, kPolDefault));
    } else {
        g.v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + VB, soff, kPolDefault));
    }
    return g;
}

template <typename T, int WIN, bool DICT, bool TAG>
__device__ __forceinline__ void sum_group_seg(T &acc, const SegGroup<T, DICT, TAG> &Q, const X4<T> &xq, const T *dict, T *ystage, uint32_t col_bits)
{
    T av[kGroupSteps];
#pragma unroll
    for (int j = 0; j < kGroupSteps; j++) av[j] = val_of<T, DICT>(Q, j, dict);
#pragma unroll
    for (int j = 0; j < kGroupSteps; j++) {
        const uint32_t cw = col_of(Q.c, j);
        acc = fma_t(av[j], x_of<T, WIN>(xq, j), acc);
        if (cw & kEndBit) {
            uint32_t row;
            if constexpr (TAG) row = j == 0 ? Q.t01 & 0xffffu : j == 1 ? Q.t01 >> 16 : j == 2 ? Q.t23 & 0xffffu : Q.t23 >> 16;
            else row = (cw & kColMask) >> col_bits;
            lds_add(ystage + row, acc);
            acc = 0;
        }
    }
}

// (Round 3's pacing of long chunks through the phases -- wavefronts of an XCD waiting for each other -- was measured with the row bands,
// lost everywhere and is gone: DESIGN.md 5.9.)
// LOADER: the workgroup has extra wavefronts (behind its `nw` computing ones) that do nothing but bring the window of x into
// LDS with LDS-direct loads (global_load_lds_dwordx4: 1 KiB per wave instruction, no registers) and leave.  The computing
// wavefronts start their loop at once, gathering everything from global memory, and meet the loaders at one s_barrier in
// front of group `gb`; from then on gathers inside the window are ds_reads.  (The window's 64 KiB take ~1.8 us to arrive --
// bandwidth, not latency -- which every wavefront used to wait out in front of its first gather; and a wavefront's own loads
// complete in order, so it cannot overlap that wait itself.)  Dictionary and zero slot are written by every computing
// wavefront for itself (the same values to the same addresses), so nothing else needs a barrier.
// PROF (CVR_DEBUG=phase_clocks; diagnostics, one extra instantiation): every wavefront stamps the 100-MHz real-time counter (s_memrealtime: one
// clock for the whole chip) at entry / prologue done / window barrier passed / loop done / rows stored, with its XCC id and the time it arrived at the window barrier:
// prof[(blockIdx.x * 16 + wave) * 8 + 0..7]; tools/phase_clocks.py turns the dump into the per-XCD histogram of profiles/.
__device__ __forceinline__ unsigned long long prof_now() { return __builtin_amdgcn_s_memrealtime(); }

template <typename T, int QA, int DEPTH, int WIN, bool DICT, bool LOADER, bool TAG, bool PROF = false>
__global__ __launch_bounds__(kLanes * kMaxWavesPerBlock) void spmv_seg_kernel(
    const uint8_t *__restrict__ stream_a, const uint4 *__restrict__ desc_a, const T *__restrict__ x, T *__restrict__ yext_a, int G, uint32_t nchunks_a,
    uint32_t nblocks_per_xcd, int swz, uint32_t cmask, uint32_t xbytes, const uint32_t *__restrict__ win_base, uint32_t wn,
    const T *__restrict__ dict_g, uint32_t ndict, uint32_t ystage_a, const uint2 *__restrict__ desc2_a, uint32_t col_bits, uint32_t nw_arg, int gb,
    const PanelArgs *__restrict__ multi, IterEpilogue epi, unsigned long long *__restrict__ prof = nullptr)
{
    unsigned long long pc[5] = {0, 0, 0, 0, 0};
    if constexpr (PROF) pc[0] = prof_now();
    const uint8_t *__restrict__ stream = stream_a;
    const uint4 *__restrict__   desc = desc_a;
    const uint2 *__restrict__   desc2 = desc2_a;
    T *__restrict__             yext = yext_a;
    uint32_t                    nchunks = nchunks_a, ystage_n = ystage_a, bidx = blockIdx.x;
    if (multi) {          // column panels, one per XCD at a time, the rounds of eight one after the other in the grid (spmv_seg_kernel)
        const uint32_t  round = blockIdx.x / nblocks_per_xcd, b = blockIdx.x - round * nblocks_per_xcd;
        const PanelArgs pa = multi[round * 8u + (b & 7u)];
        stream = pa.stream; desc = pa.desc; desc2 = pa.desc2; yext = static_cast<T *>(pa.yext); nchunks = pa.nchunks; ystage_n = pa.ystage;
        bidx = b >> 3;
    }
    constexpr int GB = (DICT ? kGroupBytesDict : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    // LDS: [waves][ystage_n] row accumulators, the value dictionary (DICT), the x window and its zero slot (WIN; wn + 4 values)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t nw = LOADER ? nw_arg : blockDim.x >> 6;                // computing wavefronts = chunks of this workgroup
    T *const ystage_all = reinterpret_cast<T *>(smem);
    T *const dict = ystage_all + nw * ystage_n;
    T *const win = dict + (DICT ? kDictMax : 0);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t blk = remap_block(bidx, nblocks_per_xcd, swz);
    const uint32_t wbase = WIN != 0 && wn && blk * nw < nchunks ? win_base[blk] : 0u;

    if constexpr (LOADER && WIN != 0) {
        if (wv >= nw) {                                                   // a loader
            constexpr uint32_t kPerLane = 16 / sizeof(T), kPerInst = kLanes * kPerLane;
            const uint32_t lw = wv - nw, nl = (blockDim.x >> 6) - nw;
            if (blk * nw < nchunks)
                for (uint32_t i0 = lw * kPerInst; i0 < wn; i0 += nl * kPerInst) {
                    const uint32_t i = i0 + lane * kPerLane;             // (wn and wbase are multiples of kPerLane: whole 16-byte pieces)
                    if (i < wn)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + wbase + i),
                                                         (__attribute__((address_space(3))) void *)(win + i0), 16, 0, 0);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (PROF) {          // (a loader: entry, its loads landed, barrier passed)
                pc[1] = prof_now();
                __builtin_amdgcn_s_barrier();
                if (lane == 0 && prof) {
                    unsigned long long *o = prof + ((size_t)blockIdx.x * 16 + wv) * 8;
                    o[0] = pc[0]; o[1] = pc[1]; o[2] = prof_now(); o[3] = 0; o[4] = 0; o[5] = __builtin_amdgcn_s_getreg(6164); o[6] = __builtin_amdgcn_s_getreg(63492); o[7] = 2;
                }
                return;
            }
            __builtin_amdgcn_s_barrier();
            return;
        }
    }

    const uint32_t k = __builtin_amdgcn_readfirstlane(blk * nw + wv);
    const bool     live = k < nchunks;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, xbytes);
    const uint32_t voff = lane * 16;
    T *const       ystage = ystage_all + wv * ystage_n;

    constexpr int  QN = DEPTH + QA;
    SegGroup<T, DICT, TAG> Q[QN];
    X4<T>          xs[DEPTH];
    // Everything the prologue loads is ISSUED before anything is waited for (the phase clocks of round 5 showed 2.9 us between entry and the
    // first gather: the accumulators were zeroed up to a count that a load brought, then the dictionary was fetched -- two memory round trips
    // one after the other behind the first stream loads): the dictionary's values into registers, the descriptors, the first groups of the
    // stream; the accumulators are zeroed up to the layout's cap, which needs no load.
    T dv[LOADER && WIN != 0 && DICT ? kDictMax / kLanes : 1];
    if constexpr (LOADER && WIN != 0 && DICT) {
#pragma unroll
        for (int u = 0; u < kDictMax / kLanes; u++) dv[u] = (uint32_t)u * kLanes + lane < ndict ? dict_g[(uint32_t)u * kLanes + lane] : T(0);
    }
    const uint4    d = live ? desc[k] : uint4{0, 0, 0, 0};
    const uint32_t nri = live ? desc2[k].y : 0u;                      // rows with a piece in this chunk
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(stream + (size_t)(live ? k : 0) * ((size_t)G * GB), live ? (uint32_t)G * GB : 0u);
#pragma unroll
    for (int i = 0; i < QN; i++) Q[i] = load_seg_group<T, DICT, TAG>(rs, voff, (uint32_t)i * GB);
    if (live) for (uint32_t i = lane; i < ystage_n; i += kLanes) ystage[i] = T(0);      // the row accumulators (+ the dump entry of the pad pieces: nri < ystage_n)

    if constexpr (LOADER && WIN != 0) {
        if constexpr (DICT) {
#pragma unroll
            for (int u = 0; u < kDictMax / kLanes; u++) dict[(uint32_t)u * kLanes + lane] = dv[u];
        }
        if (lane < 4) win[wn + lane] = T(0);
        if (epi.out && wv == 0 && lane == 0) *reinterpret_cast<uint32_t *>(win + wn + 4) = 0u;      // wavefronts that have finished (the iterative epilogue)
    } else {
        if (epi.out && threadIdx.x == 0) *reinterpret_cast<uint32_t *>(win + (WIN != 0 ? wn + 4 : 0)) = 0u;
        if constexpr (DICT)
            for (uint32_t i = threadIdx.x; i < (uint32_t)kDictMax; i += blockDim.x) dict[i] = i < ndict ? dict_g[i] : T(0);
        if constexpr (WIN != 0) {
            constexpr uint32_t kPer = 16 / sizeof(T);
            constexpr int  kBatch = 8;
            const uint32_t tstep = blockDim.x * kPer;
            for (uint32_t i0 = threadIdx.x * kPer; i0 < wn; i0 += tstep * kBatch) {
                u32x4 q[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    const uint32_t i = i0 + u * tstep;
                    q[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, i < wn ? (wbase + i) * (uint32_t)sizeof(T) : 0xfffffff0u, 0, kPolDefault));
                }
#pragma unroll
                for (int u = 0; u < kBatch; u++) if (i0 + u * tstep < wn) *reinterpret_cast<u32x4 *>(win + i0 + u * tstep) = q[u];
            }
            if (threadIdx.x < 4) win[wn + threadIdx.x] = T(0);
        }
        __syncthreads();
    }
    uint32_t wn_eff = LOADER ? 0u : wn;             // LOADER: the window is not there yet: everything through the buffer descriptor
    if constexpr (LOADER && WIN != 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // this wavefront's own dictionary / zero-slot writes before its reads
        if (!live) { __builtin_amdgcn_s_barrier(); return; }
    } else {
        if (!live) return;
    }

    double         epi_acc = 0;
    if (epi.out && epi.prev) {          // the iterative epilogue's look at the step before: its loads are in flight beside the first groups'
        constexpr int kBatch = 16;
        for (uint32_t i0 = lane; i0 < epi.nsets; i0 += kLanes * kBatch) {
            double v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) { const uint32_t i = i0 + u * kLanes; v[u] = i < epi.nsets ? epi.prev[(size_t)epi.nsets + i] : 0.0; }
#pragma unroll
            for (int u = 0; u < kBatch; u++) epi_acc += v[u];      // (ascending i: the order of a plain strided loop)
        }
    }
    T acc = 0;
    if constexpr (PROF) pc[1] = prof_now();
    // (Round 5, measured and not adopted: meeting the loaders only in front of the first group that gathers inside the window, with the column
    // phases walked from two behind the chunk's diagonal on so that the window's phase comes last -- the window's loads then share the vector
    // L1's in-order queue with the far gathers and land after 10-14 us instead of 2.5; 21.7 against 21.1 us: profiles/r05_phase_clocks_webgoogle_c_*)
    unsigned long long pc_arrive = 0;
    if constexpr (LOADER && WIN != 0) {
        if (gb < 0) { if constexpr (PROF) pc_arrive = prof_now(); asm volatile("s_barrier" ::: "memory"); wn_eff = wn; if constexpr (PROF) pc[2] = prof_now(); }      // (meet the loaders in front of the first gather)
    }
#pragma unroll
    for (int i = 0; i < DEPTH; i++) xs[i] = gather<T, kPolDefault, WIN>(rx, win, Q[i].c, cmask, wbase, wn_eff, 0u, wn);
    for (int g = 0; g < G; g++) {
        if constexpr (LOADER && WIN != 0) {
            if (g == gb) {                          // the window has arrived (the loaders waited for their loads in front of this barrier)
                if constexpr (PROF) pc_arrive = prof_now();
                asm volatile("s_barrier" ::: "memory");
                wn_eff = wn;
                if constexpr (PROF) pc[2] = prof_now();
            }
        }
        const SegGroup<T, DICT, TAG> Qn = load_seg_group<T, DICT, TAG>(rs, voff, (uint32_t)(g + QN) * GB);
        const X4<T>          xn = gather<T, kPolDefault, WIN>(rx, win, Q[DEPTH].c, cmask, wbase, wn_eff, 0u, wn);
        sum_group_seg<T, WIN, DICT, TAG>(acc, Q[0], xs[0], dict, ystage, col_bits);
#pragma unroll
        for (int i = 0; i + 1 < QN; i++) Q[i] = Q[i + 1];
        Q[QN - 1] = Qn;
#pragma unroll
        for (int i = 0; i + 1 < DEPTH; i++) xs[i] = xs[i + 1];
        xs[DEPTH - 1] = xn;
    }
    if constexpr (LOADER && WIN != 0) { if (gb >= G) { if constexpr (PROF) pc_arrive = prof_now(); asm volatile("s_barrier" ::: "memory"); if constexpr (PROF) pc[2] = prof_now(); } }      // (every wavefront meets the loaders exactly once)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pc[3] = prof_now(); }
    if (!epi.out) {
        // (y leaves past the caches -- nontemporal: on the headline's single image nobody on the chip reads it before the caller does, and its 7 MB need not
        // displace x or the window's lines in the L2s for the next SpMV: 21.05 -> 20.69 us on the web-Google shape, same box, three pairs of runs:
        // profiles/r05_headline_nt_store.log.  The same stores also write the carry slots that fixup_kernel reads next and, for column panels through this
        // kernel, the partial sums that combine_kernel reads: correct across the kernel boundary, measured on the headline only)
        for (uint32_t i = lane; i < nri; i += kLanes) {
            const uint32_t dst = i == 0 ? d.z : i == nri - 1 ? d.w : d.x + i;
            __builtin_nontemporal_store(ystage[i], yext + dst);
        }
        if constexpr (PROF) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the stores have left the wavefront; the kernel's end also waits for the L2's write-back)
            pc[4] = prof_now();
            if (lane == 0 && prof) {
                unsigned long long *o = prof + ((size_t)blockIdx.x * 16 + wv) * 8;
                o[0] = pc[0]; o[1] = pc[1]; o[2] = pc[2]; o[3] = pc[3]; o[4] = pc[4]; o[5] = __builtin_amdgcn_s_getreg(6164); o[6] = pc_arrive; o[7] = 1;
            }
        }
        return;
    }
    // The iterative caller's step in the write-out (cvr_power_iteration on a square matrix without rows cut over chunks: every
    // row is written here, once): the partial sums of x . y, y . y and x . x over this workgroup's rows, and the next iterate
    // x_next = y / ||y of the step before|| (that step's partials are summed by every wavefront itself, in one fixed order: the same
    // bits everywhere) -- instead of a pass over x and y behind every SpMV.  A wavefront's sums go to its own LDS cells; the one that
    // finishes last adds the workgroup's up in wavefront order and writes the three results to the workgroup's place.
    double inv = 1.0;
    if (epi.prev) {
        double acc = epi_acc;          // (this lane's share of the y . y partials of the step before: loaded in front of the main loop)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        inv = acc > 0 ? 1.0 / sqrt(acc) : 0.0;
    }
    T *const xnext = static_cast<T *>(epi.xnext);
    double   axy = 0, ayy = 0, axx = 0;
    constexpr int kRows = 8;            // x of eight rows per lane in flight (one round trip per 512 rows of the chunk)
    for (uint32_t i0 = lane; i0 < nri; i0 += kLanes * kRows) {
        T xv[kRows];
#pragma unroll
        for (int u = 0; u < kRows; u++) {
            const uint32_t i = i0 + u * kLanes, dst = i == 0 ? d.z : i == nri - 1 ? d.w : d.x + i;
            xv[u] = i < nri ? x[dst] : T(0);
        }
#pragma unroll
        for (int u = 0; u < kRows; u++) {
            const uint32_t i = i0 + u * kLanes;
            if (i >= nri) break;
            const uint32_t dst = i == 0 ? d.z : i == nri - 1 ? d.w : d.x + i;
            const T        yv = ystage[i];
            const double   xd = (double)xv[u], yd = (double)yv;
            store_y(yext + dst, yv);
            axy += xd * yd; ayy += yd * yd; axx += xd * xd;
            xnext[dst] = (T)(yd * inv);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { axy += __shfl_xor(axy, o); ayy += __shfl_xor(ayy, o); axx += __shfl_xor(axx, o); }
    // (this wavefront's row accumulators are free now: their first 24 bytes take its sums; the arrival counter sits behind the window)
    double   *cell = reinterpret_cast<double *>(ystage);
    uint32_t *arrived = reinterpret_cast<uint32_t *>(win + (WIN != 0 ? wn + 4 : 0));
    if (lane == 0) { cell[0] = axy; cell[1] = ayy; cell[2] = axx; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t before = 0;
    if (lane == 0) before = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    before = __builtin_amdgcn_readfirstlane(before);
    const uint32_t nlive = min(nw, nchunks - blk * nw);        // wavefronts of this workgroup with a chunk
    if (before + 1 == nlive && lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double s0 = 0, s1 = 0, s2 = 0;
        for (uint32_t w = 0; w < nlive; w++) {
            const double *cw = reinterpret_cast<const double *>(ystage_all + w * ystage_n);
            s0 += cw[0]; s1 += cw[1]; s2 += cw[2];
        }
        epi.out[blk] = s0; epi.out[(size_t)epi.nsets + blk] = s1; epi.out[2 * (size_t)epi.nsets + blk] = s2;
    }
}

// ---- interleaved images: spmv_ilv_kernel, a HAND-PIPELINED loop --------------------------------------------------------------------
// An interleaved image (cvr_ilv.hip) is the column-phase format with every slot a piece of its own, so spmv_seg_kernel runs it as it
// is; this kernel is the same arithmetic -- product rounded once, added to the row's accumulator in LDS, steps in order, lanes in order:
// the same y bit for bit -- around a loop that keeps four groups of gathers in flight per wavefront.  hipcc does not keep a ring of
// in-flight loads in fixed registers (it copies the ring's registers at the loop's back-edge behind s_waitcnt vmcnt(0): the loop of
// spmv_seg_kernel waits out a full gather round trip in every group), and with four wavefronts per CU (their accumulators fill the
// LDS) nothing else hides that latency.  So the ring lives in registers the compiler does not allocate: the kernel is held to
// v0..v95 (amdgpu_num_vgpr), stream and gather loads are issued by asm statements into v96.. and waited for with counted s_waitcnt
// vmcnt, and what a step consumes is copied out with v_mov.  Every vector-memory instruction between the run-in and the end of the loop
// is issued by these statements (one issued by the compiler there would not be counted by them).
//   x ring: D slots (the four gathered values of a group); Q ring: 2 D slots (column words, tags, values / codes of a group).
//   Step g: wait until the gathers of group g and the stream of group g + D have landed (the loads of the D - 1 steps in between stay
//   in flight), copy group g out, gather group g + D, stream group g + 2 D into the freed slots, then the arithmetic of group g.
// (The same ring around the headline kernel's loop made that kernel slower -- seven wavefronts per CU already fill the L2s' queues:
// profiles/r04_seg_ring_kernel.log.)
#ifndef CVR_RING_CAP
#define CVR_RING_CAP 40               // (tools/isa_check.py's self-test compiles with a cap the compiler cannot live in and expects the guard to refuse the result)
#endif
constexpr int kRingCap = CVR_RING_CAP;      // the compiler's registers: v0 .. v39 (it needs no more: no spill in any instantiation, checked by `make isa-check`)

template <int R> __device__ __forceinline__ void ring_ld128(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff)
{
    asm volatile("buffer_load_dwordx4 v[%0:%1], %2, %3, %4 offen" ::"n"(R), "n"(R + 3), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int R> __device__ __forceinline__ void ring_ld64(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff)
{
    asm volatile("buffer_load_dwordx2 v[%0:%1], %2, %3, %4 offen" ::"n"(R), "n"(R + 1), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int R> __device__ __forceinline__ void ring_ld32(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff)
{
    asm volatile("buffer_load_dword v[%0], %1, %2, %3 offen" ::"n"(R), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// the same for the matrix stream; NT: non-temporal -- for images too large to stay in the caches from one SpMV to the next, so that the stream's lines leave an
// L2 before its panel's slice of x does (soc-LiveJournal1 shape 256.1 -> 252.4 us, com-Orkut shape 678 -> 658; an image that does stay -- wiki-Talk -- is evicted
// by it: 33.6 -> 39.3 us; profiles/r05_stream_nt.log).  The modifier is part of the instruction: two instantiations of the kernel.
template <int R, bool NT> __device__ __forceinline__ void ring_ld128s(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff)
{
    if constexpr (NT) asm volatile("buffer_load_dwordx4 v[%0:%1], %2, %3, %4 offen nt" ::"n"(R), "n"(R + 3), "v"(voff), "s"(rs), "s"(soff) : "memory");
    else asm volatile("buffer_load_dwordx4 v[%0:%1], %2, %3, %4 offen" ::"n"(R), "n"(R + 3), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int R, bool NT> __device__ __forceinline__ void ring_ld64s(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff)
{
    if constexpr (NT) asm volatile("buffer_load_dwordx2 v[%0:%1], %2, %3, %4 offen nt" ::"n"(R), "n"(R + 1), "v"(voff), "s"(rs), "s"(soff) : "memory");
    else asm volatile("buffer_load_dwordx2 v[%0:%1], %2, %3, %4 offen" ::"n"(R), "n"(R + 1), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int R, bool NT> __device__ __forceinline__ void ring_ld32s(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff)
{
    if constexpr (NT) asm volatile("buffer_load_dword v[%0], %1, %2, %3 offen nt" ::"n"(R), "v"(voff), "s"(rs), "s"(soff) : "memory");
    else asm volatile("buffer_load_dword v[%0], %1, %2, %3 offen" ::"n"(R), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int R> __device__ __forceinline__ uint32_t ring_get()
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, v[%1]" : "=v"(v) : "n"(R) : "memory");
    return v;
}
// ring registers as OPERANDS (spmv_gang_kernel): a copy per register and group was 17 of the ~110 vector instructions of a group
template <int R> __device__ __forceinline__ uint32_t ring_and(uint32_t m)          // v[R] & m (m uniform)
{
    uint32_t v;
    asm volatile("v_and_b32 %0, %1, v[%2]" : "=v"(v) : "s"(m), "n"(R) : "memory");
    return v;
}
template <int R> __device__ __forceinline__ uint32_t ring_shr(uint32_t sh)         // v[R] >> sh (sh uniform)
{
    uint32_t v;
    asm volatile("v_lshrrev_b32 %0, %1, v[%2]" : "=v"(v) : "s"(sh), "n"(R) : "memory");
    return v;
}
template <int R, int OFF, int W> __device__ __forceinline__ uint32_t ring_bfe()    // (v[R] >> OFF) & (2^W - 1)
{
    uint32_t v;
    asm volatile("v_bfe_u32 %0, v[%1], %2, %3" : "=v"(v) : "n"(R), "n"(OFF), "n"(W) : "memory");
    return v;
}
template <int X> __device__ __forceinline__ double ring_prod(double a)             // the rounded product a * v[X : X + 1]
{
    double p;
    asm volatile("v_fma_f64 %0, %1, v[%2:%3], 0" : "=v"(p) : "v"(a), "n"(X), "n"(X + 1) : "memory");
    return p;
}
template <int X> __device__ __forceinline__ float ring_prod(float a)
{
    float p;
    asm volatile("v_fma_f32 %0, %1, v[%2], 0" : "=v"(p) : "v"(a), "n"(X) : "memory");
    return p;
}
template <int A, int X> __device__ __forceinline__ double ring_prod2(double)       // v[A : A + 1] * v[X : X + 1]
{
    double p;
    asm volatile("v_fma_f64 %0, v[%1:%2], v[%3:%4], 0" : "=v"(p) : "n"(A), "n"(A + 1), "n"(X), "n"(X + 1) : "memory");
    return p;
}
template <int A, int X> __device__ __forceinline__ float ring_prod2(float)
{
    float p;
    asm volatile("v_fma_f32 %0, v[%1], v[%2], 0" : "=v"(p) : "n"(A), "n"(X) : "memory");
    return p;
}
template <int N> __device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <typename T, bool DICT, bool TAG, int CAP = kRingCap, int XSLOTS = 4> struct RingLayout {
    static constexpr int D = 4;                                                                     // groups of gathers in flight
    static constexpr int XSZ = sizeof(T) == 8 ? 8 : 4;                                              // registers of an x slot
    static constexpr int TOFF = 4, VOFF = 4 + (TAG ? 2 : 0);                                        // tags / values inside a Q slot
    static constexpr int QSZ = (VOFF + (DICT ? 1 : sizeof(T) == 8 ? 8 : 4) + 1) & ~1;               // registers of a Q slot (even: 64-bit pairs stay aligned)
    static constexpr int NS = 1 + (TAG ? 1 : 0) + (DICT ? 1 : sizeof(T) == 8 ? 2 : 1);              // stream loads per group
    static constexpr int XB = CAP, QB = XB + XSLOTS * XSZ, TOP = QB + 2 * D * QSZ;      // (XSLOTS = D: a group's gathers land in the slot the step has just copied out; 2 D: in a slot of their own)
    static_assert(TOP <= 256, "the ring does not fit 256 registers");
    // registers of the kernel (ring included) -> wavefronts a SIMD holds -> the largest workgroup: 4 chunks' computing wavefronts + their
    // helper wavefronts (below)
    static constexpr int REGS = TOP <= 128 ? 128 : TOP <= 168 ? 168 : 256;
    static constexpr int THREADS = REGS == 128 ? 1024 : REGS == 168 ? 768 : 512;
};
template <int REGS> __device__ __forceinline__ void ring_claim()          // (the kernel's register count: the ring is invisible to the compiler)
{
    if constexpr (REGS == 128) asm volatile("" ::: "v127");
    else if constexpr (REGS == 168) asm volatile("" ::: "v167");
    else asm volatile("" ::: "v255");
}

// HELPER WAVEFRONTS (round 5).  A CU's vector L1 returns its loads in order and keeps ~100 line requests in flight; a gather that hits the L2
// holds its place for ~0.13 us, a line of the matrix stream that comes from HBM for ~1 us -- the stream is a ninth of the requests and half
// of the queue's time (the prototype with an L2-resident stream: 248 -> 179 us on the soc-LiveJournal1 shape, profiles/r05_same_stream_probe.log).
// The scalar data cache is a second path from the CU to the L2: `nw_compute` < blockDim / 64 makes the workgroup's other wavefronts HELPERS
// that touch the lines of their chunk's stream with s_load a few groups ahead of the computing wavefront's buffer loads (paced by a
// progress word the computing wavefront keeps in LDS), so that those find their lines in the L2.  Helpers read no data and write nothing.
template <typename T, bool DICT, bool TAG, bool SNT>
__global__ __launch_bounds__((RingLayout<T, DICT, TAG>::THREADS)) __attribute__((amdgpu_num_vgpr(kRingCap))) void spmv_ilv_kernel(
    const uint8_t *__restrict__ stream_a, const uint4 *__restrict__ desc_a, const uint2 *__restrict__ desc2_a, const T *__restrict__ x, T *__restrict__ yext_a, int G_alloc,
    uint32_t nchunks_a, uint32_t nblocks_per_xcd, int swz, uint32_t cmask, uint32_t xbytes_a, const T *__restrict__ dict_g, uint32_t ndict, uint32_t ystage_a, uint32_t col_bits,
    uint32_t col_base_a, const PanelArgs *__restrict__ multi, uint32_t nw_compute, uint32_t help_ahead, uint32_t help_per_line, uint32_t flip)
{
    using L = RingLayout<T, DICT, TAG>;
    constexpr int D = L::D, QN = 2 * D, XB = L::XB, QB = L::QB, K = (D - 1) * (4 + L::NS);
    ring_claim<L::REGS>();
    const uint8_t *__restrict__ stream = stream_a;
    const uint4 *__restrict__   desc = desc_a;
    const uint2 *__restrict__   desc2 = desc2_a;
    T *__restrict__             yext = yext_a;
    const uint32_t              bx = flip ? gridDim.x - 1u - blockIdx.x : blockIdx.x;          // (flip: the launch walks its workgroups backwards)
    uint32_t                    nchunks = nchunks_a, ystage_n = ystage_a, bidx = bx, col_base = col_base_a, xbytes = xbytes_a;
    if (multi) {          // column panels, one per XCD at a time (spmv_kernel); the panel's columns are relative to its first
        // (a flipped launch walks the rounds and a panel's workgroups backwards but keeps every panel on the XCD of its slot -- the workgroup's own index
        // modulo 8 --, whose L2 may still hold lines of its slice: reversing the whole index put panel s on XCD 7 - s every other SpMV)
        uint32_t       round = blockIdx.x / nblocks_per_xcd;
        const uint32_t b = blockIdx.x - round * nblocks_per_xcd;
        uint32_t       idx = b >> 3;
        if (flip) { round = gridDim.x / nblocks_per_xcd - 1u - round; idx = (nblocks_per_xcd >> 3) - 1u - idx; }
        const PanelArgs pa = multi[round * 8u + (b & 7u)];
        stream = pa.stream; desc = pa.desc; desc2 = pa.desc2; yext = static_cast<T *>(pa.yext); nchunks = pa.nchunks; ystage_n = pa.ystage;
        col_base = pa.col_base; xbytes = (pa.pad_col + 1u) * (uint32_t)sizeof(T);
        bidx = idx;
    }
    constexpr uint32_t GB = (DICT ? kGroupBytesDict : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    constexpr uint32_t VB = kColsBytes + (TAG ? kTagBytes : 0);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t nw = nw_compute, nwt = blockDim.x >> 6;           // computing wavefronts (= chunks of the workgroup), all wavefronts
    T *const ystage_all = reinterpret_cast<T *>(smem);
    T *const dict = ystage_all + nw * ystage_n;
    uint32_t *const prog = reinterpret_cast<uint32_t *>(dict + (DICT ? kDictMax : 0));      // [nw] the group each computing wavefront has reached
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t blk = remap_block(bidx, nblocks_per_xcd, swz);
    const bool     helper = wv >= nw;
    const uint32_t wc = helper ? (wv - nw) % nw : wv;                 // the computing wavefront this one is, or helps
    const uint32_t k = __builtin_amdgcn_readfirstlane(blk * nw + wc);
    const bool     live = k < nchunks;
    T *const       ystage = ystage_all + wc * ystage_n;
    constexpr uint32_t GBH = (DICT ? kGroupBytesDict : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    // PROLOGUE: everything it loads is ISSUED before anything is waited for -- the dictionary's values (into registers), the chunk's
    // descriptors, the stream of the first D groups -- and the accumulators are zeroed up to the layout's cap, which needs no load.  (It
    // used to zero up to a count that a load brought, then fetch the dictionary, then start the stream: three memory round trips one after
    // the other, ~3.4 us in front of every workgroup's first addition -- 13 % of the wiki-Talk shape's kernel.)
    T dv[4] = {T(0), T(0), T(0), T(0)};
    if constexpr (DICT) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = threadIdx.x + (uint32_t)u * blockDim.x; if (i < ndict) dv[u] = dict_g[i]; }          // (blockDim >= 64: four rounds cover the 256 entries)
    }
    const uint4    d = live ? desc[k] : uint4{0, 0, 0, 0};
    const uint2    d2 = live ? desc2[k] : uint2{0, 0};
    // (the descriptors must live in scalar registers: the asm statements below take them as such)
    const uint64_t sbase = reinterpret_cast<uint64_t>(stream + (size_t)(live ? k : 0) * ((size_t)G_alloc * GBH));
    const uint64_t sbase_u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sbase) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32)) << 32);      // (the builtin returns int: no sign extension)
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(reinterpret_cast<const void *>(sbase_u), __builtin_amdgcn_readfirstlane((uint32_t)G_alloc * GBH));      // the chunk's whole allocation: the run-in's range
    const uint32_t vo_c = lane * 16u, vo_t = lane * 8u + (uint32_t)kColsBytes, vo_code = lane * 4u + VB, vo_v0 = lane * 16u + VB, vo_v1 = vo_v0 + (uint32_t)kLanes * 16u;
    (void)vo_t; (void)vo_code; (void)vo_v1;
    // the stream of group `grp` into Q slot `qs` (a group past the chunk's last is out of range: zeros, no traffic)
    auto load_q_from = [&](const __amdgpu_buffer_rsrc_t rs, auto qsc, uint32_t grp) {
        constexpr int  R = QB + decltype(qsc)::value * L::QSZ;
        // (the group's offset goes into the VECTOR offset: the buffer's range check covers that one only, not the scalar offset -- a
        // load past the chunk's last group must not reach memory, the ring runs up to 4 D - 1 groups ahead)
        const uint32_t so = grp * GB;
        ring_ld128s<R, SNT>(vo_c + so, rs, 0u);
        if constexpr (TAG) ring_ld64s<R + L::TOFF, SNT>(vo_t + so, rs, 0u);
        if constexpr (DICT) ring_ld32s<R + L::VOFF, SNT>(vo_code + so, rs, 0u);
        else if constexpr (sizeof(T) == 8) { ring_ld128s<R + L::VOFF, SNT>(vo_v0 + so, rs, 0u); ring_ld128s<R + L::VOFF + 4, SNT>(vo_v1 + so, rs, 0u); }
        else ring_ld128s<R + L::VOFF, SNT>(vo_v0 + so, rs, 0u);
    };
    // run-in, first half: the stream of the first D groups (through the allocation's descriptor: the count of groups that hold non-zeros
    // is one of the loads in flight; groups behind it are padding of the chunk's own allocation)
    if (live && !helper) static_for<0, D>([&](auto ic) { load_q_from(rs0, ic, (uint32_t)decltype(ic)::value); });
    if (live && !helper) for (uint32_t i = lane; i < ystage_n; i += kLanes) ystage[i] = T(0);
    if (nwt > nw && threadIdx.x < nw) prog[threadIdx.x] = 0u;
    if constexpr (DICT) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = threadIdx.x + (uint32_t)u * blockDim.x; if (i < (uint32_t)kDictMax) dict[i] = dv[u]; }
        __syncthreads();
    } else {
        if (nwt > nw) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (!live) return;
    const uint32_t nri = d2.y;                                        // rows of this chunk
    const uint32_t G = __builtin_amdgcn_readfirstlane(min((uint32_t)G_alloc, d2.x));      // the groups that hold its non-zeros
    if (helper) {
        // lines of the chunk's stream, in batches of kHB per helper (what the scalar-memory counter of a wavefront tracks), helper h of H taking
        // batches h, h + H, ...; a batch is issued once the computing wavefront is within help_ahead groups of it, and skipped when that
        // wavefront's own loads have passed it
        constexpr uint32_t kHB = 15, LPG = GBH / 128u;
        static_assert(GBH % 128u == 0, "a group is whole 128-byte lines");
        const uint32_t H = (nwt - nw) / nw, h = (wv - nw) / nw;
        if (h >= H) return;
        const uint64_t hb = reinterpret_cast<uint64_t>(stream + (size_t)k * ((size_t)G_alloc * GBH));
        const uint64_t hbase = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)hb) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32)) << 32);
        const uint32_t nlines = G * LPG, last = nlines ? (nlines - 1u) * 128u : 0u;
        const uint32_t step = help_per_line > 1 ? 64u : 128u, per = help_per_line > 1 ? kHB / 2u : kHB;      // lines per batch (two loads per line: both 64-byte halves)
        for (uint32_t line = (uint32_t)(2 * QN) * LPG + h * per; line < nlines; line += H * per) {
            const uint32_t tg = line / LPG;
            uint32_t       gc;
            for (;;) {
                gc = __hip_atomic_load(prog + wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (tg <= gc + help_ahead) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (tg < gc + (uint32_t)QN + 2u) continue;                  // (too late: the computing wavefront issues the loads of group tg at group tg - QN)
            const uint32_t o0 = line * 128u;
            // (a scalar load writes its destination when the data comes back: every destination is a register of its own, distinct from the
            // operands -- early clobber -- and stays claimed until the wait, which names them all)
            uint32_t       j[kHB];
#pragma unroll
            for (uint32_t i = 0; i < kHB; i++) {
                const uint32_t off = min(o0 + i * step, last);
                asm volatile("s_load_dword %0, %1, %2" : "=&s"(j[i]) : "s"(hbase), "s"(off) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+s"(j[0]), "+s"(j[1]), "+s"(j[2]), "+s"(j[3]), "+s"(j[4]), "+s"(j[5]), "+s"(j[6]), "+s"(j[7]), "+s"(j[8]), "+s"(j[9]), "+s"(j[10]), "+s"(j[11]), "+s"(j[12]),
                           "+s"(j[13]), "+s"(j[14])
                         :: "memory");
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + col_base, xbytes);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const void *>(sbase_u), __builtin_amdgcn_readfirstlane(G * GB));
    auto load_q = [&](auto qsc, uint32_t grp) { load_q_from(rs, qsc, grp); };
    // the x of the group in Q slot `qs` into x slot `xs`
    auto gather_x = [&](auto qsc, auto xsc) {
        constexpr int R = QB + decltype(qsc)::value * L::QSZ, X = XB + decltype(xsc)::value * L::XSZ;
        const uint32_t o0 = (ring_get<R>() & cmask) * (uint32_t)sizeof(T), o1 = (ring_get<R + 1>() & cmask) * (uint32_t)sizeof(T), o2 = (ring_get<R + 2>() & cmask) * (uint32_t)sizeof(T),
                       o3 = (ring_get<R + 3>() & cmask) * (uint32_t)sizeof(T);
        if constexpr (sizeof(T) == 8) { ring_ld64<X>(o0, rx, 0u); ring_ld64<X + 2>(o1, rx, 0u); ring_ld64<X + 4>(o2, rx, 0u); ring_ld64<X + 6>(o3, rx, 0u); }
        else { ring_ld32<X>(o0, rx, 0u); ring_ld32<X + 1>(o1, rx, 0u); ring_ld32<X + 2>(o2, rx, 0u); ring_ld32<X + 3>(o3, rx, 0u); }
    };
    // run-in: the first D groups' stream, then what the steps -D .. -1 of the loop would have issued
    // (the two markers bracket the region tools/isa_check.py looks at in the compiler's output -- `make isa-check`: every vector-memory
    // instruction between them must be one of the asm statements' loads into the ring's registers, and nothing may spill)
    asm volatile("; CVR_RING_BEGIN cap=%0" ::"n"(kRingCap) : "memory");
    ring_wait<0>();
    static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        gather_x(ic, ic);
        load_q(std::integral_constant<int, i + D>{}, (uint32_t)(i + D));
    });
    for (uint32_t gb = 0; gb < G; gb += QN) {
        static_for<0, QN>([&](auto ic) {
            constexpr int  i = decltype(ic)::value, R = QB + i * L::QSZ, X = XB + (i % D) * L::XSZ;
            const uint32_t g = gb + (uint32_t)i;
            if (nwt > nw && lane == 0) prog[wv] = g;              // (helpers pace themselves by it)
            ring_wait<K>();
            const uint32_t cw[4] = {ring_get<R>(), ring_get<R + 1>(), ring_get<R + 2>(), ring_get<R + 3>()};
            uint32_t       tg[2] = {0, 0}, vv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (TAG) { tg[0] = ring_get<R + L::TOFF>(); tg[1] = ring_get<R + L::TOFF + 1>(); }
            if constexpr (DICT) vv[0] = ring_get<R + L::VOFF>();
            else static_for<0, (sizeof(T) == 8 ? 8 : 4)>([&](auto jc) { vv[decltype(jc)::value] = ring_get<R + L::VOFF + decltype(jc)::value>(); });
            static_for<0, L::XSZ>([&](auto jc) { xx[decltype(jc)::value] = ring_get<X + decltype(jc)::value>(); });
            gather_x(std::integral_constant<int, (i + D) % QN>{}, std::integral_constant<int, i % D>{});
            load_q(ic, g + (uint32_t)QN);
            if (g >= G) return;                      // (the ring runs up to 2 D - 1 groups past the chunk's last: nothing to add)
            T av[kGroupSteps];
#pragma unroll
            for (int j = 0; j < kGroupSteps; j++) {
                if constexpr (DICT) av[j] = dict[(vv[0] >> (8 * j)) & 0xffu];
                else if constexpr (sizeof(T) == 8) av[j] = __builtin_bit_cast(double, (uint64_t)vv[2 * j] | ((uint64_t)vv[2 * j + 1] << 32));
                else av[j] = __builtin_bit_cast(float, vv[j]);
            }
#pragma unroll
            for (int j = 0; j < kGroupSteps; j++) {
                T xv;
                if constexpr (sizeof(T) == 8) xv = __builtin_bit_cast(double, (uint64_t)xx[2 * j] | ((uint64_t)xx[2 * j + 1] << 32));
                else xv = __builtin_bit_cast(float, xx[j]);
                uint32_t row;
                if constexpr (TAG) row = (tg[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                else row = cw[j] >> col_bits;          // (an interleaved column word has no end flag: bits [col_bits, 32) are the row)
                lds_add(ystage + row, fma_t(av[j], xv, T(0)));          // (= the rounded product: what spmv_seg_kernel adds for a piece of one element)
            }
        });
    }
    ring_wait<0>();
    asm volatile("; CVR_RING_END" ::: "memory");
    if (nwt > nw && lane == 0) prog[wv] = 0x7ffffff0u;                // (a helper still waiting for this wavefront goes on and ends)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t i = lane; i < nri; i += kLanes) {          // the rows leave coalesced (head / last row of a chunk that shares it: its carry slot)
        const uint32_t dst = i == 0 ? d.z : i == nri - 1 ? d.w : d.x + i;
        store_y(yext + dst, ystage[i]);
    }
}


// ---- gang chunks: spmv_gang_kernel (round 6) ---------------------------------------------------------------------------------------------
// The interleaved kernel above keeps one sorted list per wavefront: a gather instruction's 64 columns come from ~36 000 non-zeros spread over
// a panel's 2-3 MB of x, about two of them per 128-byte line, and the L2s' request count -- lines, not lanes -- stays at ~0.46
// per non-zero.  A GANG is the workgroup's nw chunks sorted TOGETHER (cvr_ilv.hip): element e of the common list stands in group e / 256 of
// the gang's stream, so an instruction's 64 columns are neighbours among ~140 000 non-zeros and share their lines four times as often.  The
// wavefronts take the groups in turn, kGangUnit at a time (unit n -> wavefront n % nw); rows' sums still live in the chunks' LDS accumulators
// (a slot's tag = chunk inside the gang * ystage + row), every chunk is written out by its own wavefront as before.
// ORDER: several wavefronts now add into the same accumulators, and y must not depend on their timing.  A unit's products are computed as its
// loads land and wait in registers; the additions themselves are issued only while the unit holds the TOKEN -- a word in LDS that counts the
// units done: unit n spins until it reads n, issues its 4 kGangUnit ds_add, writes n + 1.  The LDS executes what it receives in order, so
// every row's products are added in the order of the gang's sorted list -- column order, the order of the CSR loop (spmv.cpp:1843-1850) --
// bit for bit, whatever the wavefronts do in between.  Nothing but those additions happens inside the hold (the first form held it across
// the dictionary look-ups and ran 380 us where private chunks ran 246; with only the adds inside: 205, with two groups per unit 200, and
// 178 us at 4 x 4 800 rows on the soc-LiveJournal1 shape, 510 against 736 on the com-Orkut shape: profiles/r06_token_probe_*.log).
// Column words without 16-bit tags: offset from the group's first column (17 bits) | tag (15 bits); the groups' first columns come through
// the scalar cache a revolution of the ring ahead (gbase).

// ---- the combine pass inside the panel kernel (FuseArgs, cvr_kernels.h) -----------------------------------------------------------------
// Stores and loads of the panels' partial sums go past the caches that other workgroups' stores do not reach (a CU's vector L1, another XCD's
// L2): every store of them carries sc1 (write-through), every load of them is a buffer load with sc1, and the count that hands a block over is
// an agent-scope atomic add issued behind the storing wavefronts' s_waitcnt vmcnt(0) and a workgroup barrier; the workgroup whose add returns
// the block's last count loads (MI355X_MICROARCH.md, inter-workgroup visibility: the first row of its table of measured hand-offs).
template <typename T> __device__ __forceinline__ void store_sc1(T *p, T v)
{
    typedef T __attribute__((address_space(1))) *gptr_t;
    __hip_atomic_store((gptr_t)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_store_dword(x2) ... sc1
}
template <typename T> __device__ __forceinline__ T load_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16));
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16));
}

// One block of kCombineRows rows, by all NT threads of the workgroup: y[r0 + i] = sum over the panels, in panel order, of the partial sums of row
// r0 + i (combine_kernel's arithmetic: the same bits).  lds: 12 KiB of the workgroup's LDS that nothing else uses any more.
template <typename T>
__device__ __forceinline__ void fused_combine_block(const FuseArgs &fz, uint32_t b, T *__restrict__ y, uint8_t *lds)
{
    constexpr int kBatch = 4, kEach = 4;
    T *const        acc = reinterpret_cast<T *>(lds);                                        // [kCombineRows]
    uint32_t *const s_lo = reinterpret_cast<uint32_t *>(lds + 8192), *const s_hi = s_lo + kMaxSplitPanels;
    uint64_t *const s_z = reinterpret_cast<uint64_t *>(lds + 8192 + 8 * kMaxSplitPanels), *const s_rows = s_z + kMaxSplitPanels;
    const uint32_t  NT = blockDim.x, tid = threadIdx.x, r0 = b * (uint32_t)kCombineRows, npanels = fz.npanels, nblocks = fz.nblocks;
    if (tid < npanels) {
        const CombinePanel cp = fz.panels[tid];
        s_z[tid] = reinterpret_cast<uint64_t>(cp.z); s_rows[tid] = reinterpret_cast<uint64_t>(cp.rows);
        s_lo[tid] = fz.block_off[(size_t)tid * (nblocks + 1) + b]; s_hi[tid] = fz.block_off[(size_t)tid * (nblocks + 1) + b + 1];
    }
    for (uint32_t i = tid; i < (uint32_t)kCombineRows; i += NT) acc[i] = T(0);
    __syncthreads();
    for (uint32_t p0 = 0; p0 < npanels; p0 += kBatch) {
        T        v[kBatch][kEach];
        uint32_t rw[kBatch][kEach];
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
            const uint32_t p = p0 + q;
            uint32_t       lo = 0, hi = 0;
            uint64_t       zp = 0, rp = 0;
            if (p < npanels) { lo = s_lo[p]; hi = s_hi[p]; zp = s_z[p]; rp = s_rows[p]; }
            const uint64_t zu = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)zp) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(zp >> 32)) << 32);
            const __amdgpu_buffer_rsrc_t rz = make_rsrc(reinterpret_cast<const void *>(zu), __builtin_amdgcn_readfirstlane(hi * (uint32_t)sizeof(T)));      // (entries behind hi: out of range, zeros, no traffic)
            const uint16_t *rows = reinterpret_cast<const uint16_t *>(rp);
#pragma unroll
            for (int e = 0; e < kEach; e++) {
                const uint32_t u = lo + tid + (uint32_t)e * NT;
                rw[q][e] = u < hi ? (uint32_t)rows[u] : 0xffffffffu;
                v[q][e] = load_sc1<T>(rz, u < hi ? u * (uint32_t)sizeof(T) : 0xfffffff0u);
            }
        }
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
            const uint32_t p = p0 + q;
            if (p < npanels) {                                 // (uniform)
#pragma unroll
                for (int e = 0; e < kEach; e++) if (rw[q][e] != 0xffffffffu) acc[(rw[q][e] - r0) & 0xffffu] += v[q][e];
                const uint32_t lo = s_lo[p], hi = s_hi[p];
                const uint64_t zp = s_z[p];
                const uint64_t zu = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)zp) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(zp >> 32)) << 32);
                const __amdgpu_buffer_rsrc_t rz = make_rsrc(reinterpret_cast<const void *>(zu), __builtin_amdgcn_readfirstlane(hi * (uint32_t)sizeof(T)));
                const uint16_t *rows = reinterpret_cast<const uint16_t *>(s_rows[p]);
                for (uint32_t u = lo + tid + (uint32_t)kEach * NT; u < hi; u += NT) acc[((uint32_t)rows[u] - r0) & 0xffffu] += load_sc1<T>(rz, u * (uint32_t)sizeof(T));
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < (uint32_t)kCombineRows && r0 + i < fz.nrows; i += NT) __builtin_nontemporal_store(acc[i], &y[r0 + i]);
    __syncthreads();                                           // (the next block's zeroing must not overtake these reads of acc)
}

// The gang's workgroup, once every wavefront of it has stored its rows: count in at the gang's blocks, combine those this count completed.
template <typename T>
__device__ __forceinline__ void fused_combine(const FuseArgs *__restrict__ fzp, uint32_t gang, T *__restrict__ y, uint8_t *lds)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wavefront's partial sums have left (sc1 stores)
    __syncthreads();                                           // ... and every other wavefront's
    const FuseArgs fz = *fzp;
    const uint2    rg = fz.range[gang];
    uint32_t *const list = reinterpret_cast<uint32_t *>(lds + 8192 + 24 * kMaxSplitPanels);          // [1 + 64] blocks this workgroup completed, per pass of 64
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t bb = rg.x; bb <= rg.y; bb += 64u) {
        if (wv == 0) {
            const uint32_t b = bb + lane;
            bool           last = false;
            if (b <= rg.y) last = __hip_atomic_fetch_add(fz.cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == fz.expect[b];
            const uint64_t m = __ballot(last);
            if (last) list[1 + lane_rank(m)] = b;
            if (lane == 0) list[0] = (uint32_t)__popcll(m);
        }
        __syncthreads();                                       // (the adds have returned before anybody loads)
        const uint32_t n = list[0];
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t b = list[1 + i];
            fused_combine_block<T>(fz, b, y, lds);
            if (threadIdx.x == 0) __hip_atomic_store(fz.cnt + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next SpMV
        }
        __syncthreads();
    }
}

constexpr int kGangCap = 72;          // the compiler's registers v0 .. v71 (the ilv kernel's 40 + a unit's products and tags); the ring above
template <typename T, bool DICT, bool TAG, bool SNT>
__global__ __launch_bounds__((RingLayout<T, DICT, TAG, kGangCap, 8>::THREADS)) __attribute__((amdgpu_num_vgpr(kGangCap))) void spmv_gang_kernel(
    const uint8_t *__restrict__ stream_a, const uint4 *__restrict__ desc_a, const uint2 *__restrict__ desc2_a, const T *__restrict__ x, T *__restrict__ yext_a, int G_alloc,
    uint32_t nchunks_a, uint32_t nblocks_per_xcd, int swz, uint32_t cmask, uint32_t xbytes_a, const T *__restrict__ dict_g, uint32_t ndict, uint32_t ystage_a, uint32_t col_bits,
    uint32_t col_base_a, const PanelArgs *__restrict__ multi, uint32_t nw_compute, uint32_t help_ahead, uint32_t help_per_line, uint32_t flip, const uint32_t *__restrict__ gbase_a,
    const FuseArgs *__restrict__ fuse, T *__restrict__ y_fused, uint32_t no_token)
{
    using L = RingLayout<T, DICT, TAG, kGangCap, 8>;          // (an x slot per ring position: a group's gathers never land in registers a step still reads)
    constexpr int D = L::D, QN = 2 * D, XB = L::XB, QB = L::QB, K = (D - 1) * (4 + L::NS), U = kGangUnit;
    static_assert(QN % U == 0, "a unit is whole ring slots");
    ring_claim<L::REGS>();
    const uint8_t *__restrict__  stream = stream_a;
    const uint4 *__restrict__    desc = desc_a;
    const uint2 *__restrict__    desc2 = desc2_a;
    const uint32_t *__restrict__ gbase = gbase_a;
    T *__restrict__              yext = yext_a;
    const uint32_t               bx = flip ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    uint32_t                     nchunks = nchunks_a, ystage_n = ystage_a, bidx = bx, col_base = col_base_a, xbytes = xbytes_a, gang0 = 0;
    if (multi) {          // (a flipped launch keeps every panel on its slot's XCD: spmv_ilv_kernel)
        uint32_t       round = blockIdx.x / nblocks_per_xcd;
        const uint32_t b = blockIdx.x - round * nblocks_per_xcd;
        uint32_t       idx = b >> 3;
        if (flip) { round = gridDim.x / nblocks_per_xcd - 1u - round; idx = (nblocks_per_xcd >> 3) - 1u - idx; }
        const PanelArgs pa = multi[round * 8u + (b & 7u)];
        stream = pa.stream; desc = pa.desc; desc2 = pa.desc2; yext = static_cast<T *>(pa.yext); nchunks = pa.nchunks; ystage_n = pa.ystage;
        col_base = pa.col_base; xbytes = (pa.pad_col + 1u) * (uint32_t)sizeof(T); gbase = pa.gbase; gang0 = pa.gang0;
        bidx = idx;
    }
    constexpr uint32_t GB = (DICT ? kGroupBytesDict : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    constexpr uint32_t VB = kColsBytes + (TAG ? kTagBytes : 0);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t nw = nw_compute, nwt = blockDim.x >> 6;
    T *const ystage_all = reinterpret_cast<T *>(smem);
    T *const dict = ystage_all + nw * ystage_n;
    uint32_t *const prog = reinterpret_cast<uint32_t *>(dict + (DICT ? kDictMax : 0));      // [0]: the gang's group the first wavefront has reached (the helpers' pace); [8]: the token
    uint32_t *const tok = prog + 8;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t blk = remap_block(bidx, nblocks_per_xcd, swz);
    const bool     helper = wv >= nw;
    const uint32_t kg = __builtin_amdgcn_readfirstlane(blk * nw);          // the gang's first chunk
    const uint32_t k = __builtin_amdgcn_readfirstlane(kg + (helper ? 0u : wv));      // this wavefront's own chunk: the rows it writes out
    const bool     live = kg < nchunks, own = !helper && k < nchunks;
    T dv[4] = {T(0), T(0), T(0), T(0)};
    if constexpr (DICT) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = threadIdx.x + (uint32_t)u * blockDim.x; if (i < ndict) dv[u] = dict_g[i]; }
    }
    const uint4    d = own ? desc[k] : uint4{0, 0, 0, 0};
    const uint2    d2 = own ? desc2[k] : uint2{0, 0};
    const uint32_t GGl = live ? desc2[kg].x : 0u;                        // the groups of the gang that hold its non-zeros (written by the converter)
    const uint64_t sbase = reinterpret_cast<uint64_t>(stream + (size_t)(live ? kg : 0) * ((size_t)G_alloc * GB));
    const uint64_t sbase_u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sbase) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32)) << 32);
    const uint32_t galloc = __builtin_amdgcn_readfirstlane(min(nw, nchunks - min(kg, nchunks)) * (uint32_t)G_alloc);      // groups of the gang's chunks' allocations
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(reinterpret_cast<const void *>(sbase_u), __builtin_amdgcn_readfirstlane(galloc * GB));
    const uint32_t vo_c = lane * 16u, vo_t = lane * 8u + (uint32_t)kColsBytes, vo_code = lane * 4u + VB, vo_v0 = lane * 16u + VB, vo_v1 = vo_v0 + (uint32_t)kLanes * 16u;
    (void)vo_t; (void)vo_code; (void)vo_v1;
    // the wavefront's t-th group is group gg(t) of the gang: units of U groups in turn
    const uint32_t wvU = wv * (uint32_t)U, nwU = nw * (uint32_t)U;
    auto gg = [&](uint32_t t) { return (t / (uint32_t)U) * nwU + wvU + t % (uint32_t)U; };
    auto load_q_from = [&](const __amdgpu_buffer_rsrc_t rs, auto qsc, uint32_t grp) {
        constexpr int  R = QB + decltype(qsc)::value * L::QSZ;
        const uint32_t so = grp * GB;
        ring_ld128s<R, SNT>(vo_c + so, rs, 0u);
        if constexpr (TAG) ring_ld64s<R + L::TOFF, SNT>(vo_t + so, rs, 0u);
        if constexpr (DICT) ring_ld32s<R + L::VOFF, SNT>(vo_code + so, rs, 0u);
        else if constexpr (sizeof(T) == 8) { ring_ld128s<R + L::VOFF, SNT>(vo_v0 + so, rs, 0u); ring_ld128s<R + L::VOFF + 4, SNT>(vo_v1 + so, rs, 0u); }
        else ring_ld128s<R + L::VOFF, SNT>(vo_v0 + so, rs, 0u);
    };
    if (live && !helper) static_for<0, D>([&](auto ic) { load_q_from(rs0, ic, gg((uint32_t)decltype(ic)::value)); });
    if (!helper) for (uint32_t i = lane; i < ystage_n; i += kLanes) ystage_all[wv * ystage_n + i] = T(0);
    if (threadIdx.x < 16u) prog[threadIdx.x] = 0u;
    if constexpr (DICT) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = threadIdx.x + (uint32_t)u * blockDim.x; if (i < (uint32_t)kDictMax) dict[i] = dv[u]; }
    }
    __syncthreads();                                                     // accumulators zeroed, token 0: from here on any wavefront may add into any chunk's rows
    if (!live) return;
    const uint32_t nri = d2.y;
    const uint32_t G = __builtin_amdgcn_readfirstlane(min(galloc, GGl));
    if (helper) {
        // as in spmv_ilv_kernel, over the gang's stream: helper h of all H takes batches h, h + H, ..., paced by the first wavefront's place in the gang
        constexpr uint32_t kHB = 15, LPG = GB / 128u;
        static_assert(GB % 128u == 0, "a group is whole 128-byte lines");
        const uint32_t H = nwt - nw, h = wv - nw;
        const uint32_t nlines = G * LPG, last = nlines ? (nlines - 1u) * 128u : 0u;
        const uint32_t step = help_per_line > 1 ? 64u : 128u, per = help_per_line > 1 ? kHB / 2u : kHB;
        const uint32_t ahead = help_ahead * nw, late = (uint32_t)QN * nw + 2u;
        for (uint32_t line = (uint32_t)(2 * QN) * nw * LPG + h * per; line < nlines; line += H * per) {
            const uint32_t tg = line / LPG;
            uint32_t       gc;
            for (;;) {
                gc = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (tg <= gc + ahead) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (tg < gc + late) continue;
            const uint32_t o0 = line * 128u;
            uint32_t       j[kHB];
#pragma unroll
            for (uint32_t i = 0; i < kHB; i++) {
                const uint32_t off = min(o0 + i * step, last);
                asm volatile("s_load_dword %0, %1, %2" : "=&s"(j[i]) : "s"(sbase_u), "s"(off) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+s"(j[0]), "+s"(j[1]), "+s"(j[2]), "+s"(j[3]), "+s"(j[4]), "+s"(j[5]), "+s"(j[6]), "+s"(j[7]), "+s"(j[8]), "+s"(j[9]), "+s"(j[10]), "+s"(j[11]), "+s"(j[12]),
                           "+s"(j[13]), "+s"(j[14])
                         :: "memory");
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + col_base, xbytes);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const void *>(sbase_u), __builtin_amdgcn_readfirstlane(G * GB));
    auto load_q = [&](auto qsc, uint32_t grp) { load_q_from(rs, qsc, grp); };
    // the groups' first columns (no tags): gang group g at gb[g]; a wavefront's groups of one revolution are QN / U pairs nw U apart
    // (through the scalar cache: the pointer is made wave-uniform and constant-address-space by hand -- as a generic pointer out of PanelArgs the
    // compiler loads through the vector path, which the counted waits of the ring do not allow)
    typedef const uint32_t __attribute__((address_space(4))) *cptr_t;
    const uint64_t gbp = reinterpret_cast<uint64_t>(gbase ? gbase + (size_t)kg * (size_t)G_alloc : nullptr);
    const cptr_t   gb = (cptr_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)gbp) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(gbp >> 32)) << 32));
    uint32_t bcur[QN], bnext[QN];
#pragma unroll
    for (int i = 0; i < QN; i++) { bcur[i] = 0u; bnext[i] = 0u; }
    auto bases_of = [&](uint32_t tb, uint32_t (&b)[QN]) {
        if constexpr (!TAG) {
#pragma unroll
            for (int i = 0; i < QN; i++) b[i] = gb[gg(tb + (uint32_t)i)];          // (the table has slack behind the last gang and zeros behind a gang's last group)
        }
    };
    auto gather_x = [&](auto qsc, auto xsc, uint32_t base) {
        constexpr int R = QB + decltype(qsc)::value * L::QSZ, X = XB + decltype(xsc)::value * L::XSZ;
        const uint32_t o0 = (ring_and<R>(cmask) + base) * (uint32_t)sizeof(T), o1 = (ring_and<R + 1>(cmask) + base) * (uint32_t)sizeof(T),
                       o2 = (ring_and<R + 2>(cmask) + base) * (uint32_t)sizeof(T), o3 = (ring_and<R + 3>(cmask) + base) * (uint32_t)sizeof(T);
        if constexpr (sizeof(T) == 8) { ring_ld64<X>(o0, rx, 0u); ring_ld64<X + 2>(o1, rx, 0u); ring_ld64<X + 4>(o2, rx, 0u); ring_ld64<X + 6>(o3, rx, 0u); }
        else { ring_ld32<X>(o0, rx, 0u); ring_ld32<X + 1>(o1, rx, 0u); ring_ld32<X + 2>(o2, rx, 0u); ring_ld32<X + 3>(o3, rx, 0u); }
    };
    bases_of(0u, bcur);
    // groups every wavefront walks: the same count in all of them -- every unit's turn must be taken, also one that holds nothing
    const uint32_t Tw = (G + nwU - 1u) / nwU * (uint32_t)U;
    T        prb[U][kGroupSteps];
    uint32_t rwb[U][kGroupSteps];
    asm volatile("; CVR_RING_BEGIN cap=%0" ::"n"(kGangCap) : "memory");
    ring_wait<0>();
    static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        gather_x(ic, ic, bcur[i]);
        load_q(std::integral_constant<int, i + D>{}, gg((uint32_t)(i + D)));
    });
    for (uint32_t tb = 0; tb < Tw; tb += QN) {
        bases_of(tb + (uint32_t)QN, bnext);
        static_for<0, QN>([&](auto ic) {
            constexpr int  i = decltype(ic)::value, R = QB + i * L::QSZ, X = XB + i * L::XSZ, q = i % U;
            const uint32_t t = tb + (uint32_t)i, g = gg(t);
            if (nwt > nw && wv == 0u && lane == 0) prog[0] = g;
            ring_wait<K>();
            // the gathers of the group D ahead first (into their own x slot), then what this group's stream registers hold -- tags, codes or values --, then
            // the stream of the group 2 D ahead into those registers, then the products; ring registers are operands of these instructions, not copied
            gather_x(std::integral_constant<int, (i + D) % QN>{}, std::integral_constant<int, (i + D) % QN>{}, i < D ? bcur[(i + D) % QN] : bnext[(i + D) % QN]);
            if (g < G) {          // (uniform)
                if constexpr (TAG) { rwb[q][0] = ring_bfe<R + L::TOFF, 0, 16>(); rwb[q][1] = ring_bfe<R + L::TOFF, 16, 16>(); rwb[q][2] = ring_bfe<R + L::TOFF + 1, 0, 16>(); rwb[q][3] = ring_bfe<R + L::TOFF + 1, 16, 16>(); }
                else { rwb[q][0] = ring_shr<R>(col_bits); rwb[q][1] = ring_shr<R + 1>(col_bits); rwb[q][2] = ring_shr<R + 2>(col_bits); rwb[q][3] = ring_shr<R + 3>(col_bits); }
                if constexpr (DICT) {
                    const uint32_t c0 = ring_bfe<R + L::VOFF, 0, 8>(), c1 = ring_bfe<R + L::VOFF, 8, 8>(), c2 = ring_bfe<R + L::VOFF, 16, 8>(), c3 = ring_bfe<R + L::VOFF, 24, 8>();
                    load_q(ic, gg(t + (uint32_t)QN));
                    const T a0 = dict[c0], a1 = dict[c1], a2 = dict[c2], a3 = dict[c3];
                    prb[q][0] = ring_prod<X>(a0); prb[q][1] = ring_prod<X + (int)sizeof(T) / 4>(a1); prb[q][2] = ring_prod<X + 2 * (int)sizeof(T) / 4>(a2); prb[q][3] = ring_prod<X + 3 * (int)sizeof(T) / 4>(a3);
                } else {
                    constexpr int V = R + L::VOFF, W = (int)sizeof(T) / 4;
                    prb[q][0] = ring_prod2<V, X>(T(0)); prb[q][1] = ring_prod2<V + W, X + W>(T(0)); prb[q][2] = ring_prod2<V + 2 * W, X + 2 * W>(T(0)); prb[q][3] = ring_prod2<V + 3 * W, X + 3 * W>(T(0));
                    load_q(ic, gg(t + (uint32_t)QN));
                }
            } else {          // a group behind the gang's last: +0 into the first chunk's dump entry (its loads return zeros: nothing of them is used)
                load_q(ic, gg(t + (uint32_t)QN));
#pragma unroll
                for (int j = 0; j < kGroupSteps; j++) { prb[q][j] = T(0); rwb[q][j] = ystage_n - 1u; }
            }
            if constexpr (q == U - 1) {
                if (t < Tw) {
#pragma unroll
                    for (int qq = 0; qq < U; qq++) {
#pragma unroll
                        for (int j = 0; j < kGroupSteps; j++) { asm volatile("" : "+v"(prb[qq][j])); asm volatile("" : "+v"(rwb[qq][j])); }
                    }
                    const uint32_t n = (t / (uint32_t)U) * nw + wv;     // this unit's number in the gang
                    asm volatile("" ::: "memory");
                    while (!no_token && __hip_atomic_load(tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != n) {}          // (no_token: CVR_DEBUG=gang_no_token -- the additions in whatever order, a timing experiment: y is no longer reproducible)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int qq = 0; qq < U; qq++) {
#pragma unroll
                        for (int j = 0; j < kGroupSteps; j++) lds_add_wg(ystage_all + rwb[qq][j], prb[qq][j]);
                    }
                    asm volatile("" ::: "memory");
                    if (lane == 0) __hip_atomic_store(tok, n + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                }
            }
        });
#pragma unroll
        for (int i = 0; i < QN; i++) bcur[i] = bnext[i];
    }
    ring_wait<0>();
    asm volatile("; CVR_RING_END" ::: "memory");
    if (nwt > nw && wv == 0u && lane == 0) prog[0] = 0x7ffffff0u;
    if (no_token) __syncthreads();
    if (!own && !fuse) return;
    if (own) {
        // every unit's additions are in the accumulators once the token has counted them all
        const uint32_t all = Tw / (uint32_t)U * nw;
        while (!no_token && __hip_atomic_load(tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != all) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const T *const ystage = ystage_all + wv * ystage_n;
        if (fuse) {          // the partial sums of a panel are handed to whichever workgroup completes their block: past the caches
            for (uint32_t i = lane; i < nri; i += kLanes) {
                const uint32_t dst = i == 0 ? d.z : i == nri - 1 ? d.w : d.x + i;
                store_sc1(yext + dst, ystage[i]);
            }
        } else {
            for (uint32_t i = lane; i < nri; i += kLanes) {
                const uint32_t dst = i == 0 ? d.z : i == nri - 1 ? d.w : d.x + i;
                store_y(yext + dst, ystage[i]);
            }
        }
    }
    if (fuse) fused_combine<T>(fuse, gang0 + blk, y_fused, smem);          // (no helper wavefronts with it: every wavefront of the workgroup is here)
}

// rows cut over chunks c0..c1: y[row] = carry_tail(c0) + sum_{c0 < c <= c1} carry_head(c), one wavefront per
// row, fixed summation tree (replaces the atomics of spmv.cpp:1280-1282, 1640-1649)
template <typename T>
__global__ __launch_bounds__(kLanes * kWavesPerBlock) void fixup_kernel(const int64_t *__restrict__ shared, uint32_t nshared,
                                                                        T *__restrict__ yext, uint32_t nrows)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t s = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (s >= nshared) return;
    const int64_t row = shared[3 * (size_t)s], c0 = shared[3 * (size_t)s + 1], c1 = shared[3 * (size_t)s + 2];
    const T      *carry = yext + nrows + 1;
    T             v = 0;
    for (int64_t c = c0 + 1 + lane; c <= c1; c += kLanes) v += carry[2 * c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) yext[row] = carry[2 * c0 + 1] + v;
}

// column panels: a workgroup owns kCombineRows consecutive rows.  For every panel in turn it streams the panel's
// partial sums of those rows (a contiguous range of the panel's y_ext: the panel's rows are sorted) together with their
// row numbers and adds them into LDS accumulators; a row occurs at most once per panel, and a barrier separates the
// panels, so the additions happen in panel order (bitwise reproducible).  All global accesses are coalesced streams.
// MUL: a workgroup owns MUL of those blocks of rows (block_off keeps its granularity): for matrices whose rows are mostly EMPTY -- the
// wiki-Talk shape: 6 % of the rows hold all the non-zeros -- a block of 1 024 rows has a few dozen partial sums per panel and the pass is
// all launch and round-trip latency (2 339 workgroups, 13.7 us); with MUL = 8 a workgroup has eight times the entries per round trip and
// there are an eighth of the workgroups.
template <typename T, int kBatch, int kEach, int MUL, int NT = 256>
__global__ __launch_bounds__(NT) void combine_kernel(const CombinePanel *__restrict__ panels, uint32_t npanels, const uint32_t *__restrict__ block_off,
                                                      uint32_t nblocks, T *__restrict__ y, uint32_t nrows, uint32_t plain_store)
{
    // The loads of kBatch panels are issued together (kEach entries per thread and panel in registers), then added panel by panel:
    // one memory round trip per batch instead of one per panel in front of every barrier (16 panels: 54 -> 3x us on the
    // soc-LiveJournal1 shape); what a panel holds beyond 256 * kEach entries for this block is added behind them.
    // (kBatch = 8 for up to eight panels: ONE round trip per block -- a matrix with few non-zeros per row block, the wiki-Talk shape, spends
    // its combine pass waiting for those round trips, not moving bytes)
    // Round 5: the panels' table entries (pointers, the block's range in every panel) come to LDS in one round trip in front of everything --
    // they used to be loaded inside every batch, a dependent trip in front of the data's: soc-LiveJournal1 shape 45.6 -> 36.5 us (rocprofv3
    // averages; 178 MB of partial sums, row numbers and y: 4.9 TB/s), com-Orkut shape 684 -> 672-678 us whole.  Wider batches do not add to it
    // (16 panels x 2 or x 1 entries per thread: 275 / 267 us whole against 265 with 4 x 4 -- the registers cost more occupancy than the
    // round trips they save), nor do eight blocks per workgroup outside the mostly-empty shapes (283 us): profiles/r05_combine_tables.log.
    // The row numbers are read as their low 16 bits (CombinePanel): 2 instead of 4 bytes per pair -- round 3 had found no gain in that when the pass waited
    // for round trips (r03_combine_variants.log); at 4.9 TB/s it is 23 MB less of 178 on the soc-LiveJournal1 shape (36.3 -> 35.6 us).
    constexpr uint32_t kRows = (uint32_t)kCombineRows * MUL;
    __shared__ T               acc[kRows];
    __shared__ uint32_t        s_lo[kMaxSplitPanels], s_hi[kMaxSplitPanels];
    __shared__ const T        *s_z[kMaxSplitPanels];
    __shared__ const uint16_t *s_rows[kMaxSplitPanels];
    const uint32_t b = blockIdx.x * MUL, b1 = min(b + (uint32_t)MUL, nblocks), r0 = b * kCombineRows;      // the blocks [b, b1) of the tables
    if (threadIdx.x < npanels) {
        const CombinePanel cp = panels[threadIdx.x];
        s_z[threadIdx.x] = static_cast<const T *>(cp.z); s_rows[threadIdx.x] = cp.rows;
        s_lo[threadIdx.x] = block_off[(size_t)threadIdx.x * (nblocks + 1) + b]; s_hi[threadIdx.x] = block_off[(size_t)threadIdx.x * (nblocks + 1) + b1];
    }
    for (uint32_t i = threadIdx.x; i < kRows; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < npanels; p0 += kBatch) {
        T        v[kBatch][kEach];
        uint32_t rw[kBatch][kEach];
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
            const uint32_t p = p0 + q;
            uint32_t       lo = 0, hi = 0;
            const T       *z = nullptr;
            const uint16_t *rows = nullptr;
            if (p < npanels) { lo = s_lo[p]; hi = s_hi[p]; z = s_z[p]; rows = s_rows[p]; }
#pragma unroll
            for (int e = 0; e < kEach; e++) {
                const uint32_t u = lo + threadIdx.x + (uint32_t)e * (uint32_t)NT;
                if (plain_store & 2u) {
                    rw[q][e] = u < hi ? (uint32_t)rows[u] : 0xffffffffu;
                    v[q][e] = u < hi ? z[u] : T(0);
                } else {          // (read once: past the caches -- level on the soc-LiveJournal1 and com-Orkut shapes, wiki-Talk 34.8 -> 33.8 us: profiles/r05_combine_nontemporal.log)
                    rw[q][e] = u < hi ? (uint32_t)__builtin_nontemporal_load(rows + u) : 0xffffffffu;
                    v[q][e] = u < hi ? __builtin_nontemporal_load(z + u) : T(0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
            const uint32_t p = p0 + q;
            if (p < npanels) {                                 // (uniform)
#pragma unroll
                for (int e = 0; e < kEach; e++) if (rw[q][e] != 0xffffffffu) acc[(rw[q][e] - r0) & 0xffffu] += v[q][e];
                const uint32_t  lo = s_lo[p], hi = s_hi[p];
                const T        *z = s_z[p];
                const uint16_t *rows = s_rows[p];
                for (uint32_t u = lo + threadIdx.x + (uint32_t)kEach * (uint32_t)NT; u < hi; u += (uint32_t)NT) acc[((uint32_t)rows[u] - r0) & 0xffffu] += z[u];
            }
            __syncthreads();
        }
    }
    // y is written past the caches (nontemporal): nobody on this chip reads it before the caller does, and it need not displace x or the image there
    if (plain_store & 1u) for (uint32_t i = threadIdx.x; i < kRows && r0 + i < nrows; i += blockDim.x) y[r0 + i] = acc[i];
    else for (uint32_t i = threadIdx.x; i < kRows && r0 + i < nrows; i += blockDim.x) __builtin_nontemporal_store(acc[i], &y[r0 + i]);
}

// The same pass over a BITMAP of the rows that have a partial sum in a panel (bits[(p * nblocks + b) * 32 + w]: rows b * 1 024 + 32 w ..), for matrices where
// most (row, panel) pairs hold a sum (setup_combine_bits: half or more) -- the bitmap (P bits per row) then replaces 2 bytes of row number per sum.
// A thread OWNS four rows of the block and adds their sums in registers, panel after panel: a row's sum in panel p stands at
// block_off[p][b] + (set bits of the panel below the row's), so there is nothing to scatter -- no LDS accumulators, no barrier between panels, and the loads of
// all panels of two rows are in flight together.  The additions are combine_kernel's (panel order, absent sums skipped): the same bits.
// com-Orkut shape: 40 MB of row numbers -> 3 MB of bitmap beside 160 MB of sums and 25 MB of y, 41.6 -> 33 us (profiles/r06_combine_bitmap.log); a
// load instruction is issued per (row, panel) whether the sum exists or not, which is why sparsely filled shapes keep the row numbers.
template <typename T, int PMAX>
__global__ __launch_bounds__(256) void combine_bits_kernel(const CombinePanel *__restrict__ panels, uint32_t npanels, const uint32_t *__restrict__ block_off, uint32_t nblocks,
                                                           const uint32_t *__restrict__ bits, T *__restrict__ y, uint32_t nrows, const CutEntry *__restrict__ cut, uint32_t ncut)
{
    __shared__ uint2    s_bp[PMAX][32];          // .x = the word's bits, .y = set bits of the panel in the block's words before it
    __shared__ uint32_t s_lo[PMAX];
    __shared__ const T *s_z[PMAX];
    __shared__ CutEntry s_cut[kMaxCutFold];
    const uint32_t b = blockIdx.x, r0 = b * (uint32_t)kCombineRows, tid = threadIdx.x;
    if (tid < npanels) { s_z[tid] = static_cast<const T *>(panels[tid].z); s_lo[tid] = block_off[(size_t)tid * (nblocks + 1) + b]; }
    if (tid < ncut) s_cut[tid] = cut[tid];          // (ncut = 0 for nearly every handle: no load)
    for (uint32_t idx = tid; idx < npanels * 32u; idx += 256u) {          // (a half wavefront per panel)
        const uint32_t p = idx >> 5, w = idx & 31u;
        const uint32_t v = bits[((size_t)p * nblocks + b) * 32u + w];
        uint32_t       c = (uint32_t)__popc(v);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { const uint32_t t = __shfl_up(c, o, 32); if (w >= (uint32_t)o) c += t; }
        s_bp[p][w] = uint2{v, c - (uint32_t)__popc(v)};
    }
    __syncthreads();
    // rows cut over chunks whose block this is (rare): the sum of a row's carries, by fixup_multi_kernel's instructions -- one wavefront per row, the same tree,
    // written to the same place (nobody but this workgroup reads that place)
    bool mine = false;
    for (uint32_t e = 0; e < ncut; e++) mine = mine || s_cut[e].block == b;
    if (mine) {          // (uniform)
        for (uint32_t e = tid >> 6; e < ncut; e += 4u) {
            const CutEntry c = s_cut[e];
            if (c.block != b) continue;
            const uint32_t lane = tid & 63u;
            const T       *carry = s_z[c.panel] + c.carry_off;
            T              v = 0;
            for (int64_t cc = c.c0 + 1 + lane; cc <= c.c1; cc += kLanes) v += carry[2 * cc];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            // ... stored where the panel's chunks left the row's piece (what the fix-up launch does): the loads below, behind the barrier, are this workgroup's own
            if (lane == 0) const_cast<T *>(s_z[c.panel])[c.u] = carry[2 * c.c0 + 1] + v;
        }
        __threadfence_block();
        __syncthreads();
    }
    const T *zp[PMAX];
    uint32_t lo[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; p++) {
        const bool     in = (uint32_t)p < npanels;
        const uint64_t a = reinterpret_cast<uint64_t>(in ? s_z[p] : nullptr);
        zp[p] = reinterpret_cast<const T *>((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32));
        lo[p] = __builtin_amdgcn_readfirstlane(in ? s_lo[p] : 0u);
    }
#pragma unroll
    for (int k = 0; k < kCombineRows / 256; k += 2) {
        const uint32_t iA = tid + (uint32_t)k * 256u, iB = iA + 256u, wA = iA >> 5, wB = iB >> 5, mA = 1u << (iA & 31u), mB = 1u << (iB & 31u);
        T        vA[PMAX], vB[PMAX];
        uint32_t hasA = 0, hasB = 0;
#pragma unroll
        for (int p = 0; p < PMAX; p++) {
            vA[p] = T(0); vB[p] = T(0);
            if ((uint32_t)p < npanels) {          // (uniform)
                const uint2 a = s_bp[p][wA], c = s_bp[p][wB];
                if (a.x & mA) { hasA |= 1u << p; vA[p] = __builtin_nontemporal_load(zp[p] + (lo[p] + a.y + (uint32_t)__popc(a.x & (mA - 1u)))); }
                if (c.x & mB) { hasB |= 1u << p; vB[p] = __builtin_nontemporal_load(zp[p] + (lo[p] + c.y + (uint32_t)__popc(c.x & (mB - 1u)))); }
            }
        }
        T accA = T(0), accB = T(0);
#pragma unroll
        for (int p = 0; p < PMAX; p++) {
            if (hasA & (1u << p)) accA += vA[p];
            if (hasB & (1u << p)) accB += vB[p];
        }
        if (r0 + iA < nrows) __builtin_nontemporal_store(accA, &y[r0 + iA]);
        if (r0 + iB < nrows) __builtin_nontemporal_store(accB, &y[r0 + iB]);
    }
}

// the table of the cut rows a handle's bitmap pass folds in (CutEntry, cvr_kernels.h): blockIdx.y = panel, a thread per cut row
__global__ __launch_bounds__(64) void cut_table_kernel(const FixPart *__restrict__ parts, const CombinePanel *__restrict__ panels, const uint16_t *__restrict__ rows16_base,
                                                        const uint32_t *__restrict__ rows32, CutEntry *__restrict__ out, uint32_t *__restrict__ count)
{
    const FixPart  p = parts[blockIdx.y];
    const uint32_t s = blockIdx.x * 64u + threadIdx.x;
    if (s >= p.nshared) return;
    const int64_t  row = p.shared[3 * (size_t)s], c0 = p.shared[3 * (size_t)s + 1], c1 = p.shared[3 * (size_t)s + 2];
    const uint32_t global_row = rows32[(size_t)(panels[blockIdx.y].rows - rows16_base) + (size_t)row];
    const uint32_t slot = atomicAdd(count, 1u);
    if (slot < kMaxCutFold) out[slot] = CutEntry{blockIdx.y, (uint32_t)row, global_row / (uint32_t)kCombineRows, p.nrows + 1u, c0, c1};
}

// the bitmap of one (block, panel): the low 16 bits of its sums' row numbers -> bits, through LDS
__global__ __launch_bounds__(128) void combine_bits_build_kernel(const CombinePanel *__restrict__ panels, const uint32_t *__restrict__ block_off, uint32_t nblocks, uint32_t *__restrict__ bits)
{
    __shared__ uint32_t s[32];
    const uint32_t b = blockIdx.x, p = blockIdx.y, r0 = b * (uint32_t)kCombineRows;
    if (threadIdx.x < 32u) s[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t  lo = block_off[(size_t)p * (nblocks + 1) + b], hi = block_off[(size_t)p * (nblocks + 1) + b + 1];
    const uint16_t *rows = panels[p].rows;
    for (uint32_t u = lo + threadIdx.x; u < hi; u += blockDim.x) {
        const uint32_t i = ((uint32_t)rows[u] - r0) & 0xffffu;
        atomicOr(&s[(i >> 5) & 31u], 1u << (i & 31u));
    }
    __syncthreads();
    if (threadIdx.x < 32u) bits[((size_t)p * nblocks + b) * 32u + threadIdx.x] = s[threadIdx.x];
}

// plain streaming copy: the achievable-HBM-rate yardstick beside the 8 TB/s nominal peak.  Four independent
// 16-byte loads per lane in flight, block-contiguous tiles.
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    const size_t tile = (size_t)blockDim.x * 4;
    for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
        const size_t i = base + threadIdx.x;
        u32x4        v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) if (i + (size_t)k * blockDim.x < n) v[k] = __builtin_nontemporal_load(src + i + (size_t)k * blockDim.x);
#pragma unroll
        for (int k = 0; k < 4; k++) if (i + (size_t)k * blockDim.x < n) __builtin_nontemporal_store(v[k], dst + i + (size_t)k * blockDim.x);
    }
}

// column panels: the fix-ups of all panels in one launch (blockIdx.y = panel)
template <typename T>
__global__ __launch_bounds__(kLanes * 4) void fixup_multi_kernel(const FixPart *__restrict__ parts)
{
    const FixPart  p = parts[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= p.nshared) return;
    const int64_t row = p.shared[3 * (size_t)s], c0 = p.shared[3 * (size_t)s + 1], c1 = p.shared[3 * (size_t)s + 2];
    T            *yext = static_cast<T *>(p.yext);
    const T      *carry = yext + p.nrows + 1;
    T             v = 0;
    for (int64_t c = c0 + 1 + lane; c <= c1; c += kLanes) v += carry[2 * c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) yext[row] = carry[2 * c0 + 1] + v;
}

}  // namespace

hipError_t launch_fixup_multi(const FixPart *parts, uint32_t nparts, uint32_t max_nshared, bool f32, hipStream_t st)
{
    if (nparts == 0 || max_nshared == 0) return hipSuccess;
    const dim3 grid((max_nshared + 3) / 4, nparts), block(kLanes * 4);
    if (f32) hipLaunchKernelGGL(fixup_multi_kernel<float>, grid, block, 0, st, parts);
    else hipLaunchKernelGGL(fixup_multi_kernel<double>, grid, block, 0, st, parts);
    return hipGetLastError();
}

hipError_t launch_combine_bits_build(const CombinePanel *panels, uint32_t npanels, const uint32_t *block_off, uint32_t nrows, uint32_t *bits, hipStream_t st)
{
    if (nrows == 0 || npanels == 0) return hipSuccess;
    const uint32_t nblocks = (nrows + kCombineRows - 1) / kCombineRows;
    hipLaunchKernelGGL(combine_bits_build_kernel, dim3(nblocks, npanels), dim3(128), 0, st, panels, block_off, nblocks, bits);
    return hipGetLastError();
}

hipError_t launch_cut_table(const FixPart *parts, uint32_t nparts, uint32_t max_nshared, const CombinePanel *panels, const uint16_t *rows16_base, const uint32_t *rows32, CutEntry *out, uint32_t *count, hipStream_t st)
{
    if (nparts == 0 || max_nshared == 0) return hipSuccess;
    hipLaunchKernelGGL(cut_table_kernel, dim3((max_nshared + 63) / 64, nparts), dim3(64), 0, st, parts, panels, rows16_base, rows32, out, count);
    return hipGetLastError();
}

hipError_t launch_combine(const CombinePanel *panels, uint32_t npanels, const uint32_t *block_off, void *y, uint32_t nrows, bool f32, hipStream_t st, int batch, int mul, const uint32_t *bits, const CutEntry *cut,
                          uint32_t ncut)
{
    if (nrows == 0) return hipSuccess;
    const uint32_t nblocks = (nrows + kCombineRows - 1) / kCombineRows;
    if (bits && mul == 1 && npanels <= 16u) {          // the bitmap form (combine_bits_kernel)
        if (npanels <= 8u) {
            if (f32) hipLaunchKernelGGL((combine_bits_kernel<float, 8>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, bits, static_cast<float *>(y), nrows, cut, ncut);
            else hipLaunchKernelGGL((combine_bits_kernel<double, 8>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, bits, static_cast<double *>(y), nrows, cut, ncut);
        } else {
            if (f32) hipLaunchKernelGGL((combine_bits_kernel<float, 16>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, bits, static_cast<float *>(y), nrows, cut, ncut);
            else hipLaunchKernelGGL((combine_bits_kernel<double, 16>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, bits, static_cast<double *>(y), nrows, cut, ncut);
        }
        return hipGetLastError();
    }
    const uint32_t plain = (cvr::debug_env("combine_plain_store") ? 1u : 0u) | (cvr::debug_env("combine_plain_loads") ? 2u : 0u);
    auto go = [&](auto real) {
        using T = decltype(real);
        T *yt = static_cast<T *>(y);
        // batch: 4 / 8 = that many panels per round trip with four entries per thread each; 16 / 17 = sixteen panels with two / one; 9 - 12 (eight blocks per
        // workgroup only): 1 024 or 512 threads -- 9 = 8 panels x 1 entry x 1 024 threads, 10 = 8 x 2 x 1 024, 11 = 8 x 2 x 512, 12 = 16 x 1 x 1 024
        if (mul == 8) {
            const uint32_t grid = (nblocks + 7) / 8;
            if (batch >= 16) hipLaunchKernelGGL((combine_kernel<T, 16, 2, 8>), dim3(grid), dim3(256), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else if (batch == 9) hipLaunchKernelGGL((combine_kernel<T, 8, 1, 8, 1024>), dim3(grid), dim3(1024), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);      // (1 024 threads for the eight blocks)
            else if (batch == 10) hipLaunchKernelGGL((combine_kernel<T, 8, 2, 8, 1024>), dim3(grid), dim3(1024), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else if (batch == 11) hipLaunchKernelGGL((combine_kernel<T, 8, 2, 8, 512>), dim3(grid), dim3(512), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else if (batch == 12) hipLaunchKernelGGL((combine_kernel<T, 16, 1, 8, 1024>), dim3(grid), dim3(1024), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else if (batch == 8) hipLaunchKernelGGL((combine_kernel<T, 8, 4, 8>), dim3(grid), dim3(256), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else hipLaunchKernelGGL((combine_kernel<T, 4, 4, 8>), dim3(grid), dim3(256), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
        } else {
            if (batch == 17) hipLaunchKernelGGL((combine_kernel<T, 16, 1, 1>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else if (batch == 16) hipLaunchKernelGGL((combine_kernel<T, 16, 2, 1>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else if (batch == 8) hipLaunchKernelGGL((combine_kernel<T, 8, 4, 1>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
            else hipLaunchKernelGGL((combine_kernel<T, 4, 4, 1>), dim3(nblocks), dim3(256), 0, st, panels, npanels, block_off, nblocks, yt, nrows, plain);
        }
    };
    if (f32) go(float{}); else go(double{});
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void narrow_rows_kernel(const uint32_t *__restrict__ rows, size_t n, uint16_t *__restrict__ rows16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) rows16[i] = (uint16_t)rows[i];
}
}  // namespace

hipError_t launch_narrow_rows(const uint32_t *rows, size_t n, uint16_t *rows16, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(narrow_rows_kernel, dim3((uint32_t)std::min<size_t>(4096, (n + 255) / 256)), dim3(256), 0, st, rows, n, rows16);
    return hipGetLastError();
}

hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t st)
{
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, st, static_cast<const u32x4 *>(src), static_cast<u32x4 *>(dst), n);
    return hipGetLastError();
}


namespace {
// the fused combine's tables (FuseArgs): per gang the blocks of kCombineRows rows its first and its last sub-row lie in ...
__global__ __launch_bounds__(256) void fuse_first_last_kernel(const FusePanel *__restrict__ panels, uint32_t npanels, uint32_t gw, uint32_t ngangs, uint32_t *__restrict__ first_last)
{
    const uint32_t gang = blockIdx.x * 256u + threadIdx.x;
    if (gang >= ngangs) return;
    uint32_t p = 0;
    while (p + 1 < npanels && panels[p + 1].gang0 <= gang) p++;
    const FusePanel q = panels[p];
    const uint32_t  g = gang - q.gang0, k0 = g * gw, k1 = min(k0 + gw, q.nchunks) - 1u;
    const uint32_t  s0 = q.desc[k0].x, s1 = q.desc[k1].x + max(q.desc2[k1].y, 1u) - 1u;
    first_last[2 * gang] = q.rows[s0] / (uint32_t)kCombineRows;
    first_last[2 * gang + 1] = q.rows[s1] / (uint32_t)kCombineRows;
}
// ... and from them the gang's range -- the ranges of a panel's gangs tile [0, nblocks): a gang takes the blocks up to the next gang's first, the
// first gang starts at block 0, the last ends at the last block, so that every block is completed (and its rows written, empty ones as 0) -- and
// expect[block] = the gangs of all panels whose range holds it
__global__ __launch_bounds__(256) void fuse_range_kernel(const FusePanel *__restrict__ panels, uint32_t npanels, uint32_t gw, uint32_t ngangs, uint32_t nblocks,
                                                         const uint32_t *__restrict__ first_last, uint2 *__restrict__ range, uint32_t *__restrict__ expect)
{
    const uint32_t gang = blockIdx.x * 256u + threadIdx.x;
    if (gang >= ngangs) return;
    uint32_t p = 0;
    while (p + 1 < npanels && panels[p + 1].gang0 <= gang) p++;
    const uint32_t g = gang - panels[p].gang0, ng = (panels[p].nchunks + gw - 1u) / gw;
    const uint32_t b0 = g == 0 ? 0u : first_last[2 * gang];
    const uint32_t b1 = g + 1 == ng ? nblocks - 1u : max(first_last[2 * gang + 1], first_last[2 * (gang + 1)] - min(first_last[2 * (gang + 1)], 1u));
    range[gang] = uint2{b0, b1};
    for (uint32_t b = b0; b <= b1; b++) atomicAdd(expect + b, 1u);
}
// y of the rows cut over chunks, after the fix-up launch: the partial sums of the row's sub-rows, panel by panel (the order of the combine)
template <typename T>
__global__ __launch_bounds__(64) void fuse_patch_kernel(const uint32_t *__restrict__ rows_list, uint32_t nlist, const FusePanel *__restrict__ panels, const CombinePanel *__restrict__ cpanels,
                                                         const uint32_t *__restrict__ nsub, uint32_t npanels, T *__restrict__ y)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= nlist) return;
    const uint32_t R = rows_list[i];
    T acc = T(0);
    for (uint32_t p = 0; p < npanels; p++) {
        const uint32_t *rows = panels[p].rows;
        uint32_t lo = 0, hi = nsub[p];
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rows[mid] < R) lo = mid + 1; else hi = mid; }
        if (lo < nsub[p] && rows[lo] == R) acc += static_cast<const T *>(cpanels[p].z)[lo];
    }
    y[R] = acc;
}
__global__ __launch_bounds__(64) void fuse_cut_rows_kernel(const int64_t *__restrict__ shared, uint32_t nshared, const uint32_t *__restrict__ rows, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i < nshared) out[i] = rows[shared[3 * (size_t)i]];
}
}  // namespace

hipError_t launch_fuse_setup(const FusePanel *panels_dev, uint32_t npanels, uint32_t gw, uint32_t ngangs, uint32_t nblocks, uint2 *range, uint32_t *first_last, uint32_t *expect, uint32_t *cnt, hipStream_t st)
{
    if (!ngangs || !nblocks) return hipSuccess;
    hipError_t e = hipMemsetAsync(expect, 0, sizeof(uint32_t) * nblocks, st);
    if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, sizeof(uint32_t) * nblocks, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fuse_first_last_kernel, dim3((ngangs + 255) / 256), dim3(256), 0, st, panels_dev, npanels, gw, ngangs, first_last);
    hipLaunchKernelGGL(fuse_range_kernel, dim3((ngangs + 255) / 256), dim3(256), 0, st, panels_dev, npanels, gw, ngangs, nblocks, first_last, range, expect);
    return hipGetLastError();
}

hipError_t launch_fuse_cut_rows(const int64_t *shared, uint32_t nshared, const uint32_t *rows, uint32_t *out, hipStream_t st)
{
    if (!nshared) return hipSuccess;
    hipLaunchKernelGGL(fuse_cut_rows_kernel, dim3((nshared + 63) / 64), dim3(64), 0, st, shared, nshared, rows, out);
    return hipGetLastError();
}

hipError_t launch_fuse_patch(const uint32_t *rows_list, uint32_t nlist, const FusePanel *panels_dev, const CombinePanel *cpanels, const uint32_t *nsub, uint32_t npanels, void *y, bool f32, hipStream_t st)
{
    if (!nlist) return hipSuccess;
    if (f32) hipLaunchKernelGGL(fuse_patch_kernel<float>, dim3((nlist + 63) / 64), dim3(64), 0, st, rows_list, nlist, panels_dev, cpanels, nsub, npanels, static_cast<float *>(y));
    else hipLaunchKernelGGL(fuse_patch_kernel<double>, dim3((nlist + 63) / 64), dim3(64), 0, st, rows_list, nlist, panels_dev, cpanels, nsub, npanels, static_cast<double *>(y));
    return hipGetLastError();
}

bool iter_epilogue_ok(const DeviceImage &img) { return img.phases > 1 && !img.ilv && img.nshared == 0 && img.nchunks > 0 && (img.nchunks + (img.wpb > 1 ? img.wpb : 1) - 1) / (img.wpb > 1 ? img.wpb : 1) <= 1024u; }

size_t spmv_lds_bytes(const DeviceImage &img)
{
    const uint32_t wpb = img.wpb > 1 ? img.wpb : 1;
    const bool     use_win = (img.win_elems > 0 && img.win_base != nullptr) || img.hub_n > 0;
    const uint32_t slots = img.phases > 1 ? 0u : (uint32_t)kLanes;      // (column phases: no steal slots, spmv_seg_kernel)
    return (size_t)(wpb * (slots + img.ystage) + (img.dict ? kDictMax : 0) + (use_win ? ((img.hub_n + 3u) & ~3u) + img.win_elems + 4 : 0)) * (img.f32 ? 4 : 8) + (img.phases > 1 ? 16 : 0) + (img.ilv ? 64 : 0);      // (column phases: + the arrival counter of the iterative epilogue; interleaved: + the progress words of the computing wavefronts)
}

// run-time flags -> template arguments, without macro towers: with_flag(v, f) calls f(std::true_type / false_type)
template <typename F> inline void with_flag(bool v, F &&f) { if (v) f(std::true_type{}); else f(std::false_type{}); }
template <typename F> inline void with_real(bool f32, F &&f) { if (f32) f(float{}); else f(double{}); }

hipError_t launch_spmv(const DeviceImage &img, const void *x_ext, void *y_ext, hipStream_t st, bool with_fixup, const PanelArgs *multi, uint32_t multi_chunks, uint32_t multi_rounds,
                       const IterEpilogue *epi, const FuseArgs *fuse, void *y_fused)
{
    if (img.nchunks == 0 && !multi) return hipSuccess;
    if (fuse && !(img.gang && multi && y_fused)) return hipErrorInvalidValue;
    if (epi && (multi || !iter_epilogue_ok(img))) return hipErrorInvalidValue;
    const uint32_t wpb = img.wpb > 1 ? img.wpb : 1;                     // consecutive chunks (wavefronts) per workgroup
    uint32_t       nblocks = (img.nchunks + wpb - 1) / wpb, kstride = 0;
    const size_t   lds = std::max<size_t>(spmv_lds_bytes(img), fuse ? 12288 : 0);      // (the fused combine's accumulators and tables: 12 KiB of the workgroup's LDS, whatever the chunks' own take)
    if (lds > kLdsBytes) return hipErrorInvalidValue;                   // build_part sizes the stage and the window to fit; never reached
    // a hub table without a per-workgroup window: persistent workgroups (as many as are resident at once: 16 wavefronts per CU, or what
    // the LDS allows), each staging the table once and taking chunk groups blk, blk + grid, ...  (Without a table persistent
    // wavefronts change nothing: profiles/r02_persistent_plain_layout.log.)
    if (img.hub_n && img.win_elems == 0 && img.phases <= 1 && !multi) {
        const uint32_t resident = img.ncus * std::min<uint32_t>((uint32_t)std::max<size_t>(1, kLdsBytes / std::max<size_t>(lds, 1)), std::max<uint32_t>(1u, 16u / wpb));
        if (nblocks > resident) { nblocks = resident; kstride = resident * wpb; }
    }
    const uint32_t per_xcd = (nblocks + 7) / 8;
    if (multi) multi_chunks = (multi_chunks + wpb - 1) / wpb;          // (from here on: the workgroups of wpb chunks the fullest panel needs)
    const uint32_t run8 = img.xcd_swizzle >= 3 ? 8u << (img.xcd_swizzle - 2) : 0u;       // (runs of blocks dealt over the XCDs: whole rounds of eight runs)
    const uint32_t grid = multi ? multi_rounds * multi_chunks * 8 : run8 ? (nblocks + run8 - 1) / run8 * run8 : img.xcd_swizzle == 2 ? ((per_xcd + 31) / 32) * 32 * 8 : img.xcd_swizzle ? per_xcd * 8 : nblocks;      // (multi: `img` is one of the panels: what they share comes from it)
    const uint32_t per = multi ? multi_chunks * 8 : img.xcd_swizzle == 1 || run8 ? nblocks : per_xcd;      // the kernels' block -> chunk mapping (remap_block; multi: workgroups of one round)
    const int      swz = multi ? 0 : img.xcd_swizzle;
    const uint64_t xb = (uint64_t)(img.pad_col + 1ull) * (img.f32 ? 4 : 8);
    if (xb > 0xffffffffull) return hipErrorInvalidValue;               // x is addressed through a 32-bit buffer descriptor
    const bool use_win = (img.win_elems > 0 && img.win_base != nullptr) || img.hub_n > 0, use_dict = img.dict != nullptr;
    if (img.hub_n) { const hipError_t eh = launch_hub_gather(img, x_ext, st); if (eh != hipSuccess) return eh; }
    if (img.order_n) x_ext = img.hub_x;                                 // the kernel gathers from the re-ordered copy of x
    if (img.ilv && (epi || wpb > 8u)) return hipErrorInvalidValue;      // (plan_layout keeps interleaved workgroups within eight computing wavefronts)
    // column phases with a window: four extra wavefronts per workgroup bring the window in while the others start (spmv_seg_kernel)
    const uint32_t loaders = img.phases > 1 && !img.ilv && use_win ? std::min<uint32_t>(4u, (uint32_t)kMaxWavesPerBlock - std::min<uint32_t>(wpb, kMaxWavesPerBlock)) : 0u;
    with_real(img.f32, [&](auto real) {
        using T = decltype(real);
        const T *x = static_cast<const T *>(x_ext), *dict = static_cast<const T *>(img.dict);
        T       *y = static_cast<T *>(y_ext);
        with_flag(use_dict, [&](auto DI) {
            constexpr bool kDict = decltype(DI)::value;
            if (img.ilv) {                  // interleaved chunks: the hand-pipelined kernel
                with_flag(img.tag16, [&](auto TG) {
                    // helper wavefronts (spmv_ilv_kernel): as many per chunk as the kernel's registers leave room for on the CU, at most `ilv_helpers`
                    using L = RingLayout<T, kDict, decltype(TG)::value>;
                    const uint32_t room = (uint32_t)L::THREADS / kLanes, hmax = room / wpb > 0 ? room / wpb - 1u : 0u;
                    const uint32_t H = std::min<uint32_t>(hmax, img.ilv_helpers);
                    if (img.gang) {          // gang chunks: the workgroup's wavefronts walk one common list (spmv_gang_kernel)
                        using LG = RingLayout<T, kDict, decltype(TG)::value, kGangCap, 8>;
                        const uint32_t roomg = (uint32_t)LG::THREADS / kLanes, Hg = fuse ? 0u : roomg > wpb ? std::min<uint32_t>(roomg - wpb, img.ilv_helpers * wpb) : 0u;      // helpers: all on the gang's stream (none with the fused combine: its barriers are the whole workgroup's)
                        with_flag(img.ilv_stream_nt != 0, [&](auto SN) {
                            hipLaunchKernelGGL((spmv_gang_kernel<T, kDict, decltype(TG)::value, decltype(SN)::value>), dim3(grid), dim3(kLanes * (wpb + Hg)), lds, st, img.stream, img.desc, img.desc2, x, y, img.G, img.nchunks, per, swz,
                                               img.col_mask, (uint32_t)xb, dict, img.ndict, img.ystage, img.col_bits, img.col_base, multi, wpb, img.ilv_ahead, img.ilv_per_line, img.flip_now, img.gbase, fuse, static_cast<T *>(y_fused), cvr::debug_env("gang_no_token") ? 1u : 0u);
                        });
                        return;
                    }
                    with_flag(img.ilv_stream_nt != 0, [&](auto SN) {
                        hipLaunchKernelGGL((spmv_ilv_kernel<T, kDict, decltype(TG)::value, decltype(SN)::value>), dim3(grid), dim3(kLanes * wpb * (1u + H)), lds, st, img.stream, img.desc, img.desc2, x, y, img.G, img.nchunks, per, swz,
                                           img.col_mask, (uint32_t)xb, dict, img.ndict, img.ystage, img.col_bits, img.col_base, multi, wpb, img.ilv_ahead, img.ilv_per_line, img.flip_now);
                    });
                });
            } else if (img.phases > 1) {    // column phases: every piece carries its row
                with_flag(img.tag16, [&](auto TG) { with_flag(use_win, [&](auto WI) { with_flag(loaders > 0, [&](auto LD) {
                    constexpr int kWin = decltype(WI)::value ? 1 : 0;
                    if constexpr (decltype(LD)::value && !decltype(WI)::value) return;          // (loaders only come with a window)
                    else if (img.prof && !multi && !epi && std::is_same<T, double>::value && kDict && decltype(LD)::value && !decltype(TG)::value) {
                        // the same kernel with its phases' time stamps (CVR_DEBUG=phase_clocks): that one instantiation exists -- fp64, dictionary, loader
                        // wavefronts, no 16-bit tags, the headline's --; every other layout runs its ordinary kernel (it used to launch nothing and leave y as it was)
                        if constexpr (std::is_same<T, double>::value && kDict && decltype(LD)::value && !decltype(TG)::value)
                            hipLaunchKernelGGL((spmv_seg_kernel<T, 1, 1, kWin, kDict, true, false, true>), dim3(grid), dim3(kLanes * (wpb + loaders)), lds, st, img.stream, img.desc, x, y,
                                               img.G, img.nchunks, per, swz, img.col_mask, (uint32_t)xb, img.win_base, img.win_elems, dict, img.ndict, img.ystage, img.desc2, img.col_bits, wpb, 0, multi,
                                               IterEpilogue{}, img.prof);
                    }
                    else hipLaunchKernelGGL((spmv_seg_kernel<T, 1, 1, kWin, kDict, decltype(LD)::value, decltype(TG)::value>), dim3(grid), dim3(kLanes * (wpb + loaders)), lds, st, img.stream, img.desc, x, y,
                                            img.G, img.nchunks, per, swz, img.col_mask, (uint32_t)xb, img.win_base, img.win_elems, dict, img.ndict, img.ystage, img.desc2, img.col_bits, wpb, 0, multi,
                                            epi ? *epi : IterEpilogue{});
                }); }); });
            } else if (img.c16 && !use_win && !kDict && wpb == 1 && !multi) {          // narrow chunks (banded matrices)
                if constexpr (!kDict)
                    hipLaunchKernelGGL((spmv_kernel<T, 1, kPolDefault, 1, 0, false, false, true>), dim3(grid), dim3(kLanes), lds, st, img.stream, img.desc, img.target, x, y, img.G, img.nchunks, per, swz,
                                       img.col_mask, (uint32_t)xb, img.win_base, 0u, static_cast<const T *>(nullptr), 0u, img.ystage, static_cast<const T *>(nullptr), 0u, kstride, img.cbase, img.pad_col,
                                       static_cast<const PanelArgs *>(nullptr), img.stream_mod);
            } else {                        // the general kernel: rows handed out by ballot / rank; LDS table: none, a window of x, or a hub table
                with_flag(wpb > 1, [&](auto MW) {
                    auto go = [&](auto W) {
                        hipLaunchKernelGGL((spmv_kernel<T, 1, kPolDefault, 1, decltype(W)::value, kDict, decltype(MW)::value, false>), dim3(grid), dim3(kLanes * wpb), lds, st, img.stream, img.desc, img.target,
                                           x, y, img.G, img.nchunks, per, swz, img.col_mask, (uint32_t)xb, img.win_base, img.win_elems, dict, img.ndict, img.ystage, static_cast<const T *>(img.hub_x),
                                           img.hub_n, kstride, img.cbase, img.pad_col, multi, img.stream_mod);
                    };
                    if (use_win && img.hub_n && img.ilv_stream_nt) go(std::integral_constant<int, 3>{}); else if (use_win && img.hub_n) go(std::integral_constant<int, 2>{}); else if (use_win) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 0>{});
                });
            }
        });
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || img.nshared == 0 || !with_fixup) return e;
    const uint32_t fb = (img.nshared + kWavesPerBlock - 1) / kWavesPerBlock;
    with_real(img.f32, [&](auto real) { using T = decltype(real); hipLaunchKernelGGL(fixup_kernel<T>, dim3(fb), dim3(kLanes * kWavesPerBlock), 0, st, img.shared, img.nshared, static_cast<T *>(y_ext), img.nrows); });
    return hipGetLastError();
}

// (cvr_create's warm-up thread: asking for a kernel's attributes makes the runtime load this file's code object, which the first launch would
// otherwise wait for)
void touch_spmv_kernels()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&copy_kernel));
    (void)hipGetLastError();
}

}  // namespace cvr

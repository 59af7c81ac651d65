// cvr_ilv.hip -- CSR -> CVR64 for INTERLEAVED images (cvr_options.interleave; DESIGN.md section 3, "interleaved chunks").
//
// The reference hands a lane one row at a time (spmv.cpp:821-868), so at every step the 64 lanes of a chunk gather x at the columns of
// 64 different rows: on scattered columns that is one L1->L2 request per non-zero, which is what bounds the power-law shapes
// (profiles/r03_locality_livejournal.txt: 70.1 M requests for 69.0 M non-zeros).  An interleaved chunk keeps the chunk = consecutive
// rows, equal-length lane streams, row sums in LDS, but deals its non-zeros to the lanes in COLUMN order: element e of the chunk's
// column-sorted list (ties: by row) stands at step e / 64, lane e % 64, so one gather instruction reads 64 column-sorted neighbours and
// lanes share 128-byte lines of x (the requests-per-non-zero model and the prototype: profiles/r04_request_model.log,
// r04_sorted_prototype_lj.log).  Every slot is a piece of its own: its column word carries the end flag and the row inside the chunk (or
// the 16-bit tag does), which is the column-phase format in the limit of one column per phase and pieces of one element -- the image
// runs through spmv_seg_kernel unchanged (per row the products are added in column order, as the CSR loop of spmv.cpp:1843-1850 does).
//   slots e <  n (the chunk's non-zeros)     : column | row << col_bits (bits [col_bits, 32): no end flag, every slot ends a piece), or column | end flag + tag = row; value / code
//   slots e >= n (padding up to 64 S; the pad slots the planner counts for empty rows among them): pad column (x_ext[ncols] = 0),
//                                              value 0, row = the dump entry behind the chunk's rows
// Preprocessing: one workgroup per chunk sorts the chunk's (column, position) pairs in LDS (hipCUB block radix sort, stable) and writes
// the groups.
#include "cvr_kernels.h"

#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <vector>
#include <type_traits>

namespace cvr {
namespace {

constexpr int kIlvMaxParts = 64;
// what the kernels need of one image; the images of a handle (column panels) are converted together: one sort, one pass that writes
struct IlvPartDev {
    uint8_t       *stream;
    const uint4   *desc;
    uint2         *desc2;
    const int64_t *nzb, *rp;
    const int32_t *ci;
    const void    *vals;
    int64_t        n0;          // the part's first position in its CSR arrays
    int64_t        e0;          // where its elements start in the concatenated key array
    uint32_t       nchunks, chunk0;      // its chunks are numbered chunk0 .. chunk0 + nchunks - 1 over all parts
    uint32_t       pad_col, col_base;
    uint32_t       gang0, ngangs;        // gang chunks: its gangs are numbered gang0 .. gang0 + ngangs - 1 over all parts
    uint32_t      *gbase;                // gang chunks without tags: the first column of every group
    uint32_t       ystage, pad_;         // gang chunks: accumulators per chunk of this image (a tag = chunk inside the gang * ystage + row)
};
struct IlvTable { IlvPartDev part[kIlvMaxParts]; int64_t e_end[kIlvMaxParts]; uint32_t chunk_end[kIlvMaxParts]; uint32_t nparts; };

__device__ __forceinline__ uint32_t part_of_chunk(const IlvTable *__restrict__ t, uint32_t kk)
{
    uint32_t p = 0;
    while (p + 1 < t->nparts && t->chunk_end[p] <= kk) p++;
    return p;
}

// One workgroup per chunk (its part and number are the workgroup's: no search per element) does the whole conversion on chip:
//   1. the starts of the chunk's rows (relative to its first position, clipped to the chunk) go to LDS; thread t takes the positions
//      [t IPT, (t + 1) IPT) of the chunk's CSR range and finds their rows (one search, then a walk);
//   2. a stable block radix sort (hipCUB, LDS only) of (column inside the image, position | row << 16) by the column's bits: ties keep
//      their positions' order, i.e. ascend by row; positions behind the chunk's non-zeros carry the pad column and stay behind;
//   3. element e of the sorted list = thread e % NT, item e / NT (striped), written to step e / 64, lane e % 64 of the chunk.
// (Round 4 began with one device-wide radix sort of chunk << bits | column over all panels: 4 passes of 8 bytes per non-zero through
// HBM, 2.0 of the 8.0 ms of the soc-LiveJournal1 shape's preprocessing, and 24 bytes of scratch per non-zero.)
// Also desc2[k].x = the groups of chunk k that hold non-zeros (the SpMV kernel stops there).
constexpr uint32_t kIlvF32 = 1u, kIlvDict = 2u, kIlvTag = 4u;
template <int NT, int IPT, int RB> struct ChunkSort { typedef rocprim::block_radix_sort<uint32_t, NT, IPT, uint32_t, 1, 1, (RB < 0 ? 0 : RB)> type; };      // (RB < 0: chunk_sort_to_striped instead)

// The chunk's sort, hand-written (round 5).  rocprim's block sort keeps 32 keys, 32 values and 32 ranks per thread plus its temporaries: at 1 024
// threads (128 registers) that spills 476 bytes per lane, and every stage of the conversion pays 1.5 x per element for the scratch traffic
// (profiles/r05_convert_probe.log).  Here the ranks live packed two to a register (a position inside the chunk has 16 bits), there are two
// passes of at most ten bits, and nothing else is live: a stable LSD radix sort by wavefront-wide matching --
//   items stand wave-striped (wavefront w, item i, lane l <-> sorted index w * 64 IPT + 64 i + l); per pass every item finds the lanes of its
//   wavefront with the same digit (one ballot per bit), the lowest of them adds their number to the wavefront's counter of that digit (LDS) and
//   passes the old value on: rank among the wavefront's items = old value + peers in lower lanes; a scan over (digit, wavefront) turns the
//   counters into first positions; keys and values go to their positions through LDS (one array at a time: 33 words per 32, no bank conflicts
//   on the way back) and are read back wave-striped again, after the last pass striped over the workgroup (element e = item e / 1 024 of thread
//   e % 1 024: what the write-out expects).
// In: key / val blocked (thread t holds positions t IPT .. t IPT + IPT - 1).  cbits <= 20.
// Measured (soc-LiveJournal1 shape, 444-step chunks, us per chunk; CVR_DEBUG=ilv_clocks): to wave-striped 3.7 | pass 1: ranks 24, scan 3, positions 2, moves 16 |
// pass 2: 24, 4, 3, 18 = 97 for the sort, as rocprim's (97) -- the compiler still spills 240 bytes per lane around it (476 with rocprim; 380 before the
// digits and the padded addresses were made opaque between their uses, which kept 32 + 32 registers alive across the barriers) -- but the stage in front of
// it falls from 55 to 24 us and the write-out from 51 to 40: the launch 2.52 -> 2.00 ms (CVR_DEBUG=ilv_rocprim_sort: the library sort).  The ranking is not
// bound by its LDS atomics (a plain read and write per peer group instead: the same 24 us) nor by the ballots' branches (unrolled: the same) nor by
// spills (half the values parked in LDS meanwhile: the same 24 us, not kept): ~110 vector instructions per item at 4 cycles each for a 64-wide wavefront, 32 items, four
// wavefronts per SIMD = 27 us.
template <int IPT>
__device__ __forceinline__ void chunk_sort_to_striped(uint32_t (&key)[IPT], uint32_t (&val)[IPT], uint8_t *smem, uint32_t cbits, unsigned long long *tclk = nullptr)
{
    int stamp_i = 0;
    auto stamp = [&]() { if (tclk && threadIdx.x == 0) tclk[stamp_i] = __builtin_amdgcn_s_memrealtime(); stamp_i++; };
    constexpr uint32_t NT = 1024, NW = 16, WI = (uint32_t)IPT * 64u;
    uint32_t *const xch = reinterpret_cast<uint32_t *>(smem);            // [NT IPT 33 / 32]; the counters [NW][2^bits] alias it
    uint32_t *const cnt = xch;
    uint32_t *const wsum = xch + std::max<uint32_t>(NT * (uint32_t)IPT / 32u * 33u, NW * 1024u);          // [NW], behind both (dynamic LDS: the kernel asks for all of the CU's)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    auto pad = [](uint32_t i) { return i + (i >> 5); };
    auto blocked_to_wave_striped = [&](uint32_t (&a)[IPT]) {
#pragma unroll
        for (int i = 0; i < IPT; i++) xch[pad(tid * (uint32_t)IPT + (uint32_t)i)] = a[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IPT; i++) a[i] = xch[pad(w * WI + (uint32_t)i * 64u + lane)];
        __syncthreads();
    };
    stamp();
    blocked_to_wave_striped(key);
    blocked_to_wave_striped(val);
    stamp();
    auto pass = [&](uint32_t shift, uint32_t bits, bool last) {
        const uint32_t ND = 1u << bits, dm = ND - 1u;
        for (uint32_t j = tid; j < NW * ND; j += NT) cnt[j] = 0;
        __syncthreads();
        uint32_t rk[IPT / 2];
#pragma unroll
        for (int i = 0; i < IPT / 2; i++) rk[i] = 0;
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const uint32_t d = (key[i] >> shift) & dm;
            unsigned long long m = ~0ull;
            for (uint32_t bb = 0; bb < bits; bb++) { const bool one = (d >> bb) & 1u; const unsigned long long bal = __ballot(one); m &= one ? bal : ~bal; }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)), total = (uint32_t)__popcll(m);
            uint32_t base = 0;
            // (the counters of a wavefront are its own, and the lowest lanes of the peer groups have different digits: a plain read and write --
            // LDS operations of a wavefront execute in order -- where an atomic with return cost ~90 cycles per item and wavefront: 23 -> x us per pass)
            if (lower == 0) { volatile uint32_t *cw = cnt + w * ND + d; base = *cw; *cw = base + total; }
            base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
            rk[i >> 1] |= (base + lower) << (16 * (i & 1));
        }
        __syncthreads();
        stamp();
        uint32_t run = 0;
        if (tid < ND) for (uint32_t v = 0; v < NW; v++) { const uint32_t c = cnt[v * ND + tid]; cnt[v * ND + tid] = run; run += c; }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += u; }
        if (lane == 63u) wsum[w] = incl;
        __syncthreads();
        uint32_t first = incl - run;
        for (uint32_t v = 0; v < w; v++) first += wsum[v];
        if (tid < ND) for (uint32_t v = 0; v < NW; v++) cnt[v * ND + tid] += first;
        __syncthreads();
        stamp();
        uint32_t shift2 = shift;
        asm volatile("" : "+s"(shift2));          // (opaque: the digits are computed again here, not kept in 32 registers since the ranking above)
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const uint32_t d = (key[i] >> shift2) & dm, sh = 16u * ((uint32_t)i & 1u), lr = (rk[i >> 1] >> sh) & 0xffffu;
            const uint32_t pos = cnt[w * ND + d] + lr;
            rk[i >> 1] = (rk[i >> 1] & ~(0xffffu << sh)) | (pos << sh);
        }
        __syncthreads();
        stamp();
        auto move = [&](uint32_t (&a)[IPT]) {
            uint32_t zero = 0;
            asm volatile("" : "+v"(zero));          // (opaque: the 32 padded addresses are computed per array, not kept from one array's move to the other's)
#pragma unroll
            for (int i = 0; i < IPT; i++) xch[pad(((rk[i >> 1] >> (16 * (i & 1))) & 0xffffu) + zero)] = a[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < IPT; i++) a[i] = xch[pad((last ? (uint32_t)i * NT + tid : w * WI + (uint32_t)i * 64u + lane) + zero)];
            __syncthreads();
        };
        move(val);
        move(key);
        stamp();
    };
    const uint32_t b0 = (cbits + 1u) / 2u;
    pass(0u, b0, false);
    pass(b0, cbits - b0, true);
}

template <int NT, int IPT, int RB>
__global__ __launch_bounds__(NT) void ilv_chunk_kernel(const IlvTable *__restrict__ t, const void *__restrict__ dict_v, uint32_t ndict, int G, uint32_t col_bits, uint32_t cbits,
                                                                  uint32_t flags, uint32_t *__restrict__ err, unsigned long long *__restrict__ clk)
{
    typedef typename ChunkSort<NT, IPT, RB>::type Sort;
    if (clk && threadIdx.x == 0) clk[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime();          // (CVR_DEBUG=ilv_clocks: the stages of a chunk on the 100-MHz counter)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *const   rstart = reinterpret_cast<uint32_t *>(smem);       // [nri]; the sort's storage takes its place afterwards
    const bool        f32 = flags & kIlvF32, use_dict = flags & kIlvDict, tag = flags & kIlvTag;
    const uint32_t    kk = blockIdx.x;
    const IlvPartDev &q = t->part[part_of_chunk(t, kk)];
    const uint32_t    k = kk - q.chunk0;
    const int64_t     b = q.nzb[k];
    const uint32_t    n = (uint32_t)(q.nzb[k + 1] - b);
    const uint32_t    row_first = q.desc[k].x, nri = q.desc2[k].y;
    if (threadIdx.x == 0) q.desc2[k].x = (n + 255u) / 256u;
    for (uint32_t i = threadIdx.x; i < nri; i += blockDim.x) {
        const int64_t r = q.rp[row_first + i] - b;                       // (a row cut over chunks begins before the chunk)
        rstart[i] = (uint32_t)(r < 0 ? 0 : r > (int64_t)n ? (int64_t)n : r);
    }
    __syncthreads();
    uint32_t       key[IPT], val[IPT];
    const uint32_t p0 = threadIdx.x * IPT;
    uint32_t       row = 0;
    if (p0 < n && nri > 1) {                                              // the last row that starts at or before p0
        uint32_t lo = 0, hi = nri;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rstart[mid] <= p0) lo = mid; else hi = mid; }
        row = lo;
    }
#pragma unroll
    for (int i = 0; i < IPT; i++) {
        const uint32_t p = p0 + i;
        if (p < n) {
            while (row + 1 < nri && rstart[row + 1] <= p) row++;
            key[i] = (uint32_t)q.ci[b + p] - q.col_base;
            val[i] = p | (row << 16);
        } else { key[i] = q.pad_col; val[i] = 0xffffffffu; }
    }
    __syncthreads();
    if (clk && threadIdx.x == 0) clk[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (RB < 0) chunk_sort_to_striped<IPT>(key, val, smem, cbits, clk ? clk + (size_t)blockIdx.x * 16 + 4 : nullptr);
    else Sort().sort_to_striped(key, val, *reinterpret_cast<typename Sort::storage_type *>(smem), 0u, cbits);
    if (clk && threadIdx.x == 0) clk[blockIdx.x * 16 + 2] = __builtin_amdgcn_s_memrealtime();

    // the dictionary (sorted by bit pattern, at most 256 entries) goes to LDS in the sort's place: a search per element is eight LDS reads
    uint64_t *const dl = reinterpret_cast<uint64_t *>(smem);
    uint32_t        code0 = 0;                                            // the code of +0.0 (cvr_create puts it into every dictionary)
    if (use_dict) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 256u; i += NT)
            dl[i] = i < ndict ? (f32 ? (uint64_t)static_cast<const uint32_t *>(dict_v)[i] : static_cast<const uint64_t *>(dict_v)[i]) : ~(uint64_t)0;
        __syncthreads();
        while (code0 < ndict && dl[code0] != 0) code0++;
        if (code0 >= ndict) { if (threadIdx.x == 0) atomicOr(err, 4u); code0 = 0; }
    }
    const uint32_t GB = (use_dict ? kGroupBytesDict : f32 ? kGroupBytes32 : kGroupBytes64) + (tag ? kTagBytes : 0);
    const uint32_t VB = kColsBytes + (tag ? kTagBytes : 0);
    const uint32_t nslots = (uint32_t)G * 256u;
    uint8_t *const base = q.stream + (size_t)k * G * GB;
    const void *const vals = q.vals;
    const uint32_t pad_col = q.pad_col;
    constexpr int  kB = IPT % 8 == 0 ? 8 : 4;                             // the values' bit patterns: the loads of kB elements in flight together
#pragma unroll
    for (int i0 = 0; i0 < IPT; i0 += kB) {
        uint64_t bits[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) {
            const uint32_t e = (uint32_t)(i0 + u) * NT + threadIdx.x, pe = val[i0 + u] & 0xffffu;      // pe: the element's place among the chunk's, in CSR order
            bits[u] = e >= n ? 0 : f32 ? (uint64_t)static_cast<const uint32_t *>(vals)[b + pe] : static_cast<const uint64_t *>(vals)[b + pe];
        }
#pragma unroll
        for (int u = 0; u < kB; u++) {
            const int      i = i0 + u;
            const uint32_t e = (uint32_t)i * NT + threadIdx.x;
            if (e < nslots) {
                const uint32_t g = e >> 8, j = (e >> 6) & 3u, lane = e & 63u;
                const uint32_t col = e < n ? key[i] : pad_col, r = e < n ? val[i] >> 16 : nri;
                uint32_t       code = code0;
                if (use_dict && e < n) {
                    uint32_t a = 0;                                       // entries below bits[u] (branch-free: the entries behind ndict are all ones)
#pragma unroll
                    for (uint32_t st = 128; st > 0; st >>= 1) a += dl[a + st - 1] < bits[u] ? st : 0u;
                    if (a >= ndict || dl[a] != bits[u]) { atomicOr(err, 4u); a = 0; }
                    code = a;
                }
                uint8_t *grp = base + (size_t)g * GB;
                const uint32_t cw = tag ? col | kEndBit : col | (r << col_bits);          // (no end flag without tags: every slot ends a piece, the row takes bits [col_bits, 32))
                reinterpret_cast<uint32_t *>(grp)[lane * 4 + j] = cw;
                if (tag) reinterpret_cast<uint16_t *>(grp + kColsBytes)[lane * 4 + j] = (uint16_t)r;
                if (use_dict) (grp + VB)[lane * 4 + j] = (uint8_t)code;
                else if (!f32) reinterpret_cast<uint64_t *>(grp + VB + (j >> 1) * (kLanes * 16))[lane * 2 + (j & 1)] = bits[u];
                else reinterpret_cast<uint32_t *>(grp + VB)[lane * 4 + j] = (uint32_t)bits[u];
            }
        }
    }
    if (clk && threadIdx.x == 0) clk[blockIdx.x * 16 + 3] = __builtin_amdgcn_s_memrealtime();
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
inline uint32_t bits_of(uint64_t v) { uint32_t b = 1; while (b < 63 && ((uint64_t)1 << b) <= v) b++; return b; }      // bits that hold 0 .. v

// ---- gang chunks (cvr_format.h; spmv_gang_kernel) -----------------------------------------------------------------------------------------
// The gw chunks of a workgroup are sorted TOGETHER: the list of a gang holds up to gw * 64 S non-zeros (147 000 at four chunks of 576 steps),
// more than a workgroup sorts in LDS -- so all gangs of all images go through ONE device-wide stable radix sort of (gang << cbits | column,
// position), four or five passes of eight bits over 8 bytes per non-zero (rocPRIM), and one workgroup per gang writes the gang's groups:
//   1. gang_key_kernel   : per chunk, key = gang number << cbits | column inside the image, value = the element's place in the images' concatenated order
//   2. radix_sort_pairs  : stable -- equal (gang, column) keep their positions' order, i.e. ascend by row
//   3. gang_write_kernel : per gang: the chunks' row starts to LDS; item (group, lane) = the four slots of a lane in a group: chunk and row of each
//                          element's position (compare with the chunks' first positions, search among the chunk's rows) -> tag = chunk * ystage + row;
//                          column word = column - the group's first column | tag << 17 (gbase[group] = that column), or column | end flag + 16-bit tag;
//                          value or dictionary code; desc2[first chunk].x = the gang's groups that hold non-zeros
// A column further than 2^17 from its group's first sets *err_flag bit 3: cvr_preprocess converts again with 16-bit tags (sparse panels).
template <typename K>
__global__ __launch_bounds__(256) void gang_key_kernel(const IlvTable *__restrict__ t, uint32_t gw, uint32_t cbits, K *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        int64_t *__restrict__ gang_e0, uint32_t ngangs_tot, int64_t n_tot)
{
    const uint32_t    kk = blockIdx.x;
    const IlvPartDev &q = t->part[part_of_chunk(t, kk)];
    const uint32_t    k = kk - q.chunk0;
    const int64_t     b = q.nzb[k], e = q.nzb[k + 1];
    const uint32_t    gang = q.gang0 + k / gw;
    const K           hi = (K)gang << cbits;
    if (threadIdx.x == 0 && k % gw == 0) gang_e0[gang] = q.e0 + (b - q.n0);
    if (kk == 0 && threadIdx.x == 0) gang_e0[ngangs_tot] = n_tot;
    for (int64_t p = b + threadIdx.x; p < e; p += 256) {
        const int64_t i = q.e0 + (p - q.n0);
        keys[i] = hi | (K)((uint32_t)q.ci[p] - q.col_base);
        vals[i] = (uint32_t)i;
    }
}

template <typename K>
__global__ __launch_bounds__(1024) void gang_write_kernel(const IlvTable *__restrict__ t, const K *__restrict__ keys, const uint32_t *__restrict__ vals, const int64_t *__restrict__ gang_e0,
                                                           const void *__restrict__ dict_v, uint32_t ndict, int G, uint32_t gw, uint32_t cbits, uint32_t flags,
                                                           uint32_t *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *const dl = reinterpret_cast<uint64_t *>(smem);                   // [256] the dictionary
    uint32_t *const rstart = reinterpret_cast<uint32_t *>(smem + 2048);        // [gw][ystage] the chunks' row starts, relative to the gang's first position
    __shared__ int64_t  s_cb[kMaxWavesPerBlock + 1];                           // the chunks' first positions (relative likewise), [nc] = the gang's end
    __shared__ uint32_t s_nri[kMaxWavesPerBlock];
    const bool     f32 = flags & kIlvF32, use_dict = flags & kIlvDict, tag = flags & kIlvTag;
    const uint32_t gi = blockIdx.x;
    uint32_t       pi = 0;
    while (pi + 1 < t->nparts && t->part[pi].gang0 + t->part[pi].ngangs <= gi) pi++;
    const IlvPartDev &q = t->part[pi];
    const uint32_t    ystage = q.ystage;
    const uint32_t    kg = (gi - q.gang0) * gw, nc = min(gw, q.nchunks - kg);
    const int64_t     p0 = q.nzb[kg];                                          // the gang's first position in the part's CSR arrays
    const int64_t     E0 = gang_e0[gi];
    const uint32_t    n = (uint32_t)(gang_e0[gi + 1] - E0);
    const uint32_t    GGn = (n + 255u) / 256u;
    if (threadIdx.x <= nc) s_cb[threadIdx.x] = q.nzb[kg + threadIdx.x] - p0;
    if (threadIdx.x < nc) { s_nri[threadIdx.x] = q.desc2[kg + threadIdx.x].y; q.desc2[kg + threadIdx.x].x = threadIdx.x == 0 ? GGn : 0u; }
    uint32_t code0 = 0;
    if (use_dict) for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) dl[i] = i < ndict ? (f32 ? (uint64_t)static_cast<const uint32_t *>(dict_v)[i] : static_cast<const uint64_t *>(dict_v)[i]) : ~(uint64_t)0;
    __syncthreads();
    for (uint32_t c = 0; c < nc; c++) {
        const uint32_t row_first = q.desc[kg + c].x, nri = s_nri[c];
        const int64_t  cb = s_cb[c], ce = s_cb[c + 1];
        for (uint32_t i = threadIdx.x; i < nri; i += blockDim.x) {
            const int64_t r = q.rp[row_first + i] - p0;                        // (a row cut over chunks begins before its chunk)
            rstart[c * ystage + i] = (uint32_t)(r < cb ? cb : r > ce ? ce : r);
        }
    }
    __syncthreads();
    if (use_dict) {
        while (code0 < ndict && dl[code0] != 0) code0++;
        if (code0 >= ndict) { if (threadIdx.x == 0) atomicOr(err, 4u); code0 = 0; }
    }
    const uint32_t GB = (use_dict ? kGroupBytesDict : f32 ? kGroupBytes32 : kGroupBytes64) + (tag ? kTagBytes : 0);
    const uint32_t VB = kColsBytes + (tag ? kTagBytes : 0);
    uint8_t *const base = q.stream + (size_t)kg * G * GB;
    const K        cm = ((K)1 << cbits) - 1;
    const uint32_t dump = s_nri[0];                                            // the first chunk's dump entry: where the padding behind the gang's last element adds its zeros
    const void *const va = q.vals;
    for (uint32_t it = threadIdx.x; it < GGn * 64u; it += blockDim.x) {
        const uint32_t g = it >> 6, lane = it & 63u;
        const uint32_t bcol = (uint32_t)(keys[E0 + (int64_t)g * 256] & cm);    // the group's first = smallest column
        uint32_t cw[4], tg[4], code[4];
        uint64_t bits[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t el = g * 256u + (uint32_t)j * 64u + lane;
            cw[j] = tag ? q.pad_col | kEndBit : 0u | (dump << kGangOffBits);
            tg[j] = dump; code[j] = code0; bits[j] = 0;
            if (el < n) {
                const uint32_t col = (uint32_t)(keys[E0 + el] & cm);
                const int64_t  rel = (int64_t)vals[E0 + el] - q.e0 + q.n0 - p0;    // the element's position, relative to the gang's first
                uint32_t       c = 0;
                while (c + 1 < nc && s_cb[c + 1] <= rel) c++;
                const uint32_t *rs = rstart + c * ystage;
                uint32_t        lo = 0, hi = s_nri[c];                         // the last row of the chunk that starts at or before the position
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)rs[mid] <= rel) lo = mid; else hi = mid; }
                tg[j] = c * ystage + lo;
                if (tag) cw[j] = col | kEndBit;
                else {
                    const uint32_t off = col - bcol;
                    if (off >> kGangOffBits) atomicOr(err, 8u);
                    cw[j] = (off & ((1u << kGangOffBits) - 1u)) | (tg[j] << kGangOffBits);
                }
                const int64_t p = p0 + rel;
                bits[j] = f32 ? (uint64_t)static_cast<const uint32_t *>(va)[p] : static_cast<const uint64_t *>(va)[p];
            }
        }
        if (use_dict) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t el = g * 256u + (uint32_t)j * 64u + lane;
                if (el < n) {
                    uint32_t a = 0;
#pragma unroll
                    for (uint32_t st = 128; st > 0; st >>= 1) a += dl[a + st - 1] < bits[j] ? st : 0u;
                    if (a >= ndict || dl[a] != bits[j]) { atomicOr(err, 4u); a = 0; }
                    code[j] = a;
                }
            }
        }
        uint8_t *grp = base + (size_t)g * GB;
        reinterpret_cast<uint4 *>(grp)[lane] = uint4{cw[0], cw[1], cw[2], cw[3]};
        if (tag) reinterpret_cast<uint2 *>(grp + kColsBytes)[lane] = uint2{tg[0] | (tg[1] << 16), tg[2] | (tg[3] << 16)};
        if (use_dict) reinterpret_cast<uint32_t *>(grp + VB)[lane] = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
        else if (!f32) {
            reinterpret_cast<uint64_t *>(grp + VB)[lane * 2] = bits[0]; reinterpret_cast<uint64_t *>(grp + VB)[lane * 2 + 1] = bits[1];
            reinterpret_cast<uint64_t *>(grp + VB + kLanes * 16)[lane * 2] = bits[2]; reinterpret_cast<uint64_t *>(grp + VB + kLanes * 16)[lane * 2 + 1] = bits[3];
        } else reinterpret_cast<uint4 *>(grp + VB)[lane] = uint4{(uint32_t)bits[0], (uint32_t)bits[1], (uint32_t)bits[2], (uint32_t)bits[3]};
        if (!tag && lane == 0 && q.gbase) q.gbase[(size_t)kg * G + g] = bcol;
    }
}

template <typename K>
hipError_t convert_gang_typed(const IlvTable &tab, IlvTable *d_tab, uint32_t nchunks_tot, uint32_t ngangs_tot, int64_t N, const DeviceImage &c0, uint32_t ystage_max, uint32_t cbits, uint32_t gbits,
                              uint32_t *err_flag, uint8_t *scratch, size_t scratch_bytes, size_t off, hipStream_t st)
{
    (void)tab;
    const size_t n = (size_t)N;
    K        *k_in = reinterpret_cast<K *>(scratch + off);            off += up256(sizeof(K) * n);
    K        *k_out = reinterpret_cast<K *>(scratch + off);           off += up256(sizeof(K) * n);
    uint32_t *v_in = reinterpret_cast<uint32_t *>(scratch + off);     off += up256(sizeof(uint32_t) * n);
    uint32_t *v_out = reinterpret_cast<uint32_t *>(scratch + off);    off += up256(sizeof(uint32_t) * n);
    int64_t  *ge0 = reinterpret_cast<int64_t *>(scratch + off);       off += up256(sizeof(int64_t) * ((size_t)ngangs_tot + 1));
    if (off > scratch_bytes) return hipErrorInvalidValue;
    size_t temp_bytes = scratch_bytes - off;
    size_t need = 0;
    hipError_t e = n ? rocprim::radix_sort_pairs(nullptr, need, k_in, k_out, v_in, v_out, n, 0u, cbits + gbits, st) : hipSuccess;
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gang_key_kernel<K>, dim3(nchunks_tot), dim3(256), 0, st, d_tab, c0.gang, cbits, k_in, v_in, ge0, ngangs_tot, N);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (n) e = rocprim::radix_sort_pairs(scratch + off, need, k_in, k_out, v_in, v_out, n, 0u, cbits + gbits, st);
    if (e != hipSuccess) return e;
    const uint32_t flags = (c0.f32 ? kIlvF32 : 0u) | (c0.dict ? kIlvDict : 0u) | (c0.tag16 ? kIlvTag : 0u);
    const size_t   lds = 2048 + sizeof(uint32_t) * (size_t)c0.gang * ystage_max;
    static bool    attr[2] = {false, false};
    if (!attr[sizeof(K) == 8]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gang_write_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsBytes - 1024));
        if (e != hipSuccess) return e;
        attr[sizeof(K) == 8] = true;
    }
    if (lds > kLdsBytes - 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gang_write_kernel<K>, dim3(ngangs_tot), dim3(1024), lds, st, d_tab, k_out, v_out, ge0, c0.dict, c0.ndict, c0.G, c0.gang, cbits, flags, err_flag);
    return hipGetLastError();
}

template <int NT, int IPT, int RB = 0>
hipError_t launch_chunks(const IlvTable *d_tab, uint32_t nchunks_tot, const DeviceImage &c, uint32_t ystage_max, uint32_t cbits, uint32_t *err_flag, hipStream_t st)
{
    const size_t   sort_lds = RB < 0 ? std::max<size_t>((size_t)NT * IPT / 32 * 33 * sizeof(uint32_t), (size_t)16 * 1024 * sizeof(uint32_t)) + 256 : sizeof(typename ChunkSort<NT, IPT, RB>::type::storage_type);
    const size_t   lds = std::max(sort_lds, std::max<size_t>(2048, sizeof(uint32_t) * (size_t)std::max<uint32_t>(ystage_max, 1u)));
    const uint32_t flags = (c.f32 ? kIlvF32 : 0u) | (c.dict ? kIlvDict : 0u) | (c.tag16 ? kIlvTag : 0u);
    static bool    attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ilv_chunk_kernel<NT, IPT, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (lds > kLdsBytes) return hipErrorInvalidValue;
    unsigned long long *clk = nullptr;
    if (cvr::debug_env("ilv_clocks") && hipMalloc(&clk, sizeof(unsigned long long) * 16 * (size_t)nchunks_tot) != hipSuccess) { (void)hipGetLastError(); clk = nullptr; }
    hipLaunchKernelGGL((ilv_chunk_kernel<NT, IPT, RB>), dim3(nchunks_tot), dim3(NT), lds, st, d_tab, c.dict, c.ndict, c.G, c.col_bits, cbits, flags, err_flag, clk);
    hipError_t le = hipGetLastError();
    if (clk) {          // diagnostics: mean time per stage over the chunks, and the launch from its first stamp to its last
        std::vector<unsigned long long> hc(16 * (size_t)nchunks_tot);
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hc.data(), clk, sizeof(unsigned long long) * hc.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            double d[3] = {0, 0, 0}, ds[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            unsigned long long t0 = ~0ull, t1 = 0;
            for (uint32_t k = 0; k < nchunks_tot; k++) {
                for (int i = 0; i < 3; i++) d[i] += (double)(hc[16 * k + i + 1] - hc[16 * k + i]);
                t0 = std::min(t0, hc[16 * k]); t1 = std::max(t1, hc[16 * k + 3]);
                if (RB < 0) for (int i = 0; i < 9; i++) ds[i] += (double)(hc[16 * k + 5 + i] - hc[16 * k + 4 + i]);
            }
            fprintf(stderr, "[ilv_clocks] %u chunks x %d threads x %d pairs (%s), LDS %zu: load + rows %.1f us, sort %.1f us, dictionary + write %.1f us per chunk; launch %.1f us\n", nchunks_tot, NT, IPT, RB < 0 ? "own sort" : "rocprim", lds,
                    d[0] / nchunks_tot / 100.0, d[1] / nchunks_tot / 100.0, d[2] / nchunks_tot / 100.0, (double)(t1 - t0) / 100.0);
            if (RB < 0) fprintf(stderr, "[ilv_clocks] own sort: to wave-striped %.1f | pass 1: ranks %.1f scan %.1f positions %.1f moves %.1f | pass 2: %.1f %.1f %.1f %.1f us\n", ds[0] / nchunks_tot / 100, ds[1] / nchunks_tot / 100,
                            ds[2] / nchunks_tot / 100, ds[3] / nchunks_tot / 100, ds[4] / nchunks_tot / 100, ds[5] / nchunks_tot / 100, ds[6] / nchunks_tot / 100, ds[7] / nchunks_tot / 100, ds[8] / nchunks_tot / 100);
        }
        (void)hipFree(clk);
    }
    return le;
}
}  // namespace

// device scratch of a conversion (the table of the images; the sort itself happens in LDS)
size_t convert_interleaved_scratch(int64_t nnz, uint32_t nchunks, bool gang)
{
    size_t bytes = up256(sizeof(IlvTable)) + 256;
    if (gang) {          // keys (64 bits at most) and positions, in and out; the gangs' first elements (a gang has two chunks or more); the sort's own storage
        const size_t n = (size_t)std::max<int64_t>(nnz, 0);
        size_t       temp = 0;
        uint64_t    *k = nullptr;
        uint32_t    *v = nullptr;
        if (n && rocprim::radix_sort_pairs(nullptr, temp, k, k, v, v, n, 0u, 64u, (hipStream_t) nullptr) != hipSuccess) temp = n * 16 + (1 << 20);
        bytes += 2 * up256(8 * n) + 2 * up256(4 * n) + up256(8 * ((size_t)nchunks + 2 * kIlvMaxParts + 2)) + up256(temp) + 4096;
    }
    return bytes;
}

// The interleaved images imgs[0 .. n) (all of one handle: same chunk length, value type, dictionary, tag width and row field) converted
// together -- one sort over all their non-zeros, one pass that writes all their groups; csrs[i], [n0[i], n1[i]) = image i's CSR arrays and
// positions.  scratch: convert_interleaved_scratch(sum of the non-zeros, ..) bytes.  *err_flag bit 2: a value that is not in the dictionary.
hipError_t launch_convert_interleaved(const DeviceImage *const *imgs, const DeviceCsr *csrs, const int64_t *n0, const int64_t *n1, int n, uint32_t *err_flag, void *scratch,
                                      size_t scratch_bytes, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    if (n > kIlvMaxParts) return hipErrorInvalidValue;
    IlvTable tab;
    memset(&tab, 0, sizeof(tab));
    int64_t  e = 0;
    uint32_t c = 0, pad_max = 0, ystage_max = 0;
    for (int i = 0; i < n; i++) {
        const DeviceImage &g = *imgs[i];
        if (g.G != imgs[0]->G || g.f32 != imgs[0]->f32 || g.dict != imgs[0]->dict || g.tag16 != imgs[0]->tag16 || g.col_bits != imgs[0]->col_bits) return hipErrorInvalidValue;
        IlvPartDev &q = tab.part[i];
        q.stream = g.stream; q.desc = g.desc; q.desc2 = g.desc2; q.nzb = csrs[i].nz_begin; q.rp = csrs[i].row_ptr; q.ci = csrs[i].col_idx; q.vals = csrs[i].vals;
        q.n0 = n0[i]; q.e0 = e; q.nchunks = g.nchunks; q.chunk0 = c; q.pad_col = g.pad_col; q.col_base = g.col_base;
        e += n1[i] - n0[i]; c += g.nchunks;
        tab.e_end[i] = e; tab.chunk_end[i] = c;
        pad_max = std::max(pad_max, g.pad_col);
        ystage_max = std::max(ystage_max, g.ystage);
    }
    tab.nparts = (uint32_t)n;
    if (c == 0) return hipSuccess;
    const DeviceImage &c0 = *imgs[0];
    const uint32_t     cbits = bits_of(pad_max);
    const int          ipt = (c0.S + 15) / 16;                             // 64 S slots over 1024 threads
    static_assert((kIlvMaxSteps + 15) / 16 <= 36, "the longest interleaved chunk needs more than 36 pairs per thread");
    if (up256(sizeof(IlvTable)) > scratch_bytes || ipt > 36 || ystage_max > 65536u) return hipErrorInvalidValue;      // (position 16 bits, row 16 bits in the sort's payload; 36 pairs per thread: 144 KiB of LDS)
    IlvTable *d_tab = static_cast<IlvTable *>(scratch);
    if (c0.gang) {          // gang chunks: one device-wide sort, one workgroup per gang writes
        uint32_t ng = 0;
        for (int i = 0; i < n; i++) {
            if (imgs[i]->gang != c0.gang || (!c0.tag16 && !imgs[i]->gbase)) return hipErrorInvalidValue;
            tab.part[i].gang0 = ng; tab.part[i].ngangs = (imgs[i]->nchunks + c0.gang - 1) / c0.gang; tab.part[i].gbase = imgs[i]->gbase; tab.part[i].ystage = imgs[i]->ystage;
            ng += tab.part[i].ngangs;
        }
        if (e >= (int64_t)0xffffffffll || c0.gang > (uint32_t)kMaxWavesPerBlock || (uint64_t)c0.gang * ystage_max > (c0.tag16 ? 65536ull : (1ull << kGangTagBits))) return hipErrorInvalidValue;
        hipError_t rg = hipMemcpyAsync(d_tab, &tab, sizeof(IlvTable), hipMemcpyHostToDevice, st);
        if (rg != hipSuccess) return rg;
        const uint32_t gbits = bits_of(ng ? ng - 1 : 0);
        uint8_t *const sc = static_cast<uint8_t *>(scratch);
        if (cbits + gbits <= 32) return convert_gang_typed<uint32_t>(tab, d_tab, c, ng, e, c0, ystage_max, cbits, gbits, err_flag, sc, scratch_bytes, up256(sizeof(IlvTable)), st);
        return convert_gang_typed<uint64_t>(tab, d_tab, c, ng, e, c0, ystage_max, cbits, gbits, err_flag, sc, scratch_bytes, up256(sizeof(IlvTable)), st);
    }
    hipError_t rc = hipMemcpyAsync(d_tab, &tab, sizeof(IlvTable), hipMemcpyHostToDevice, st);
    if (rc != hipSuccess) return rc;
    // (1 024 threads leave 128 registers each: 16 (column, position) pairs per thread sort without spills, 24 spill 92 bytes, 32 spill 470;
    // 512 threads x 48 / 64 pairs spill more and ran slower: 2.8 against 2.2 ms on the soc-LiveJournal1 shape)
    if (cbits <= 20 && !cvr::debug_env("ilv_rocprim_sort")) {          // the hand-written sort (chunk_sort_to_striped): no spills at any length
        if (ipt <= 4) return launch_chunks<1024, 4, -1>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
        if (ipt <= 8) return launch_chunks<1024, 8, -1>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
        if (ipt <= 12) return launch_chunks<1024, 12, -1>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
        if (ipt <= 16) return launch_chunks<1024, 16, -1>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
        if (ipt <= 24) return launch_chunks<1024, 24, -1>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
        if (ipt <= 32) return launch_chunks<1024, 32, -1>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
        return launch_chunks<1024, 36, -1>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
    }
    if (ipt <= 4) return launch_chunks<1024, 4>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
    if (ipt <= 8) return launch_chunks<1024, 8>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
    if (ipt <= 12) return launch_chunks<1024, 12>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
    if (ipt <= 16) return launch_chunks<1024, 16>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
    if (ipt <= 24) return launch_chunks<1024, 24>(d_tab, c, c0, ystage_max, cbits, err_flag, st);
    // 32 / 36 pairs per thread: digits of ten bits -- the 19-20 bits of a panel's columns in two passes instead of three of eight: the sort of a
    // 444-step chunk 124 -> 98 us, the launch 2.66 -> 2.52 ms on the soc-LiveJournal1 shape (six bits: 145 us; profiles/r05_convert_probe.log).
    // What these lengths cost beside 16 pairs (no spills, 64 KiB of LDS, two workgroups per CU: 1.48 ms for the same matrix at S = 256) is the
    // price of the long chunks the SpMV wants (305 us at S = 256 against 275).
    if (ipt <= 32 && cvr::debug_env("ilv_rb8")) return launch_chunks<1024, 32>(d_tab, c, c0, ystage_max, cbits, err_flag, st);      // (diagnostics: the former three passes)
    if (ipt <= 32) return launch_chunks<1024, 32, 10>(d_tab, c, c0, ystage_max, cbits, err_flag, st);          // (768 threads x 44 pairs -- 170 registers each -- spill 364 bytes and take as long)
    return launch_chunks<1024, 36, 10>(d_tab, c, c0, ystage_max, cbits, err_flag, st);          // (chunks of up to 576 steps: what lets a launch of 8 200 chunks at S = 508 fit eight generations of workgroups instead of nine)
}

}  // namespace cvr

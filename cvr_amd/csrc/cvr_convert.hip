// cvr_convert.hip -- CSR -> CVR64 on the device, one wavefront per chunk (gfx950, wave64).
//
// This is the reference's tracker loop (pre_processing, /root/reference/spmv.cpp:711-1000) with the
// 8 AVX-512 trackers (valID, rowID, count; spmv.cpp:711-759) widened to 64 lanes:
//   * "is any tracker empty" (_mm512_mask_reduce_min_epi32, spmv.cpp:810) is one __ballot;
//   * the scalar in-lane-order refill loop (spmv.cpp:814-946) becomes rank-among-empty-lanes
//     (mbcnt of the ballot): lane with rank r takes segment fed + r -- the identical assignment;
//   * stealing (spmv.cpp:869-943: ave = remaining steps, victim = FIRST lane with count > ave,
//     the stealer takes the FIRST ave elements of the victim's remainder) is resolved victim by
//     victim in a wave-uniform loop: a victim with count c serves ceil(c/ave)-1 consecutive stealers;
//   * the 8-wide CSR gathers + 64-B stores (spmv.cpp:963-977) are per-lane loads and one 16-B store
//     per lane per 4 steps, already in the layout the SpMV kernel streams (cvr_format.h).
// No (pos, wb) records are written (spmv.cpp:832-834, 898-899): the end of a segment is bit 31 of
// the column word, and the row a lane writes follows from the hand-out order.
#include "cvr_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace cvr {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_dw __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte load from a dword-aligned address
typedef double   f64x2 __attribute__((ext_vector_type(2)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <typename T> struct Bits;
template <> struct Bits<double> { typedef uint64_t type; };
template <> struct Bits<float>  { typedef uint32_t type; };

// code of value v in the dictionary (LDS, sorted by bit pattern; the host put every value of the matrix and +0.0 in)
template <typename T>
__device__ __forceinline__ uint32_t dict_code(const typename Bits<T>::type *dict, uint32_t ndict, T v)
{
    const typename Bits<T>::type b = __builtin_bit_cast(typename Bits<T>::type, v);
    uint32_t lo = 0, hi = ndict;            // first entry >= b
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (dict[mid] < b) lo = mid + 1; else hi = mid; }
    return lo;
}

// CVR_CONVERT_CLOCKS: 100-MHz time stamps of every 256th chunk's stages (launch_convert prints them)
__device__ unsigned long long *g_conv_dbg = nullptr;
#define CONV_CLOCK(i) do { if (g_conv_dbg && lane == 0 && (k & 255u) == 0) g_conv_dbg[(k >> 8) * 16 + (i)] = wall_clock64(); } while (0)

// SEGT: the chunk's segments come from a table (column phases: one segment per (row, phase) pair, launch_seg_build) instead
// of being the chunk's rows in order.
// STAGE: the wavefront first copies its chunk's feed table -- the (begin, length, row) of its segments, or its clamped row
// pointers -- into LDS with coalesced loads.  The hand-out of a new segment then costs an LDS read instead of a global load that
// every later load of the step depends on: the converter is a chain of 64 S dependent steps, 75 % of its cycles are waits
// (SQ_WAIT_ANY), and this removes one of the two round trips per step.  (Staging the chunk's columns and values as well was
// slower: 34 KB of LDS per chunk leave four chunks per CU where seven want to run; profiles/r02_convert_lds_staging_probe.log.)
template <typename T, bool DICT, bool SEGT, bool C16, bool STAGE = false, bool TAG = false>
__global__ __launch_bounds__(kLanes * kWavesPerBlock) void convert_kernel(
    const int64_t *__restrict__ rp, const int32_t *__restrict__ cidx, const T *__restrict__ vals,
    const int64_t *__restrict__ nzb, const uint32_t *__restrict__ pad_cnt, const uint4 *__restrict__ desc,
    uint8_t *__restrict__ stream, uint8_t *__restrict__ target, uint32_t *__restrict__ err, int G,
    uint32_t nchunks, uint32_t pad_col, const T *__restrict__ dict_g, uint32_t ndict,
    const uint2 *__restrict__ desc2, const int64_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_len,
    const uint16_t *__restrict__ seg_row, uint32_t col_bits, const uint32_t *__restrict__ seg_flags, const int32_t *__restrict__ hub_index, const uint32_t *__restrict__ hub_bitmap,
    const uint32_t *__restrict__ cbase, uint32_t hub_n, uint32_t stage_bytes, uint32_t seg_packed, const uint32_t *__restrict__ nchunks_dev)
{
    if (nchunks_dev) nchunks = min(nchunks, *nchunks_dev);      // (a plan beyond the launch's room: the host falls back)
    extern __shared__ __attribute__((aligned(16))) uint8_t csm[];      // STAGE: stage_bytes per wavefront
    if constexpr (SEGT) { if (seg_flags[0] & 3u) return; }      // unsorted rows: the segment table is meaningless (cvr_preprocess reports it)
    constexpr int GB = (DICT ? kGroupBytesDict : C16 ? (sizeof(T) == 8 ? kGroupBytes64C16 : kGroupBytes32C16) : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    constexpr int CB = (C16 ? kCols16Bytes : kColsBytes) + (TAG ? kTagBytes : 0);      // bytes of the group in front of the values: column words (+ wide row tags)
    typedef typename Bits<T>::type bits_t;
    __shared__ bits_t dict[DICT ? kDictMax : 1];
    if constexpr (DICT) {
        for (uint32_t i = threadIdx.x; i < ndict; i += blockDim.x) dict[i] = __builtin_bit_cast(bits_t, dict_g[i]);
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (k >= nchunks) return;
    CONV_CLOCK(0);
    const int64_t  b = nzb[k], e = nzb[k + 1];
    const uint4    d = desc[k];
    const uint32_t row_first = d.x, nseg = d.y, pc = pad_cnt[k];
    const uint32_t nrow_seg = nseg - (pc > 0 ? 1u : 0u);
    const int      S = G * kGroupSteps;
    uint32_t       sbase = 0;
    if constexpr (SEGT) sbase = desc2[k].x;

    int64_t  pos = -1;     // CSR element this lane emits next; -1 = pad slot
    uint32_t cnt = 0;      // slots left in the lane's current segment
    uint32_t fed = 0;      // segments handed out so far (wave-uniform)
    uint32_t tgt = lane;   // lane this one stole from
    uint32_t rowtag = 0;   // SEGT: the chunk's row of the lane's segment, shifted above the column index (goes into its last column word)
    uint32_t bad = 0;
    uint8_t *out = stream + (size_t)k * G * GB + lane * (C16 ? 8 : 16);
    uint32_t base_col = 0;
    if constexpr (C16) base_col = cbase[k];

    // STAGE, per wavefront: SEGT: begin (u16, from the chunk's first element; 0xffff = pad), length (u16), row (u16) x 64 S;
    // else the row pointers of the chunk's rows, clamped to the chunk, from its first element (u32) x (64 S + 2).  Addressed as
    // offsets into csm, so that every access is an LDS instruction.
    const uint32_t cap = (uint32_t)S * kLanes, o_feed = (threadIdx.x >> 6) * stage_bytes;
#define SBEG(i) (*reinterpret_cast<uint16_t *>(csm + o_feed + 2u * (uint32_t)(i)))
#define SLEN(i) (*reinterpret_cast<uint16_t *>(csm + o_feed + 2u * cap + 2u * (uint32_t)(i)))
#define SRW(i)  (*reinterpret_cast<uint16_t *>(csm + o_feed + 4u * cap + 2u * (uint32_t)(i)))
#define SROW(i) (*reinterpret_cast<uint32_t *>(csm + o_feed + 4u * (uint32_t)(i)))
    if constexpr (STAGE) {
        constexpr uint32_t kBatch = 8;              // loads in flight per lane: the copy is a few round trips, not one per 64 entries
        if constexpr (SEGT) {
            if (seg_packed) {       // 8-byte records {begin | length << 16, row}: one load per segment
                const uint2 *pk = reinterpret_cast<const uint2 *>(seg_begin) + sbase;
                for (uint32_t q0 = 0; q0 < nseg; q0 += kLanes * kBatch) {
                    uint2 r[kBatch];
#pragma unroll
                    for (uint32_t u = 0; u < kBatch; u++) {
                        const uint32_t q = q0 + u * kLanes + lane;
                        r[u] = q < nseg ? pk[q] : uint2{0u, 0u};
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kBatch; u++) {
                        const uint32_t q = q0 + u * kLanes + lane;
                        if (q < nseg) { SBEG(q) = (uint16_t)r[u].x; SLEN(q) = (uint16_t)(r[u].x >> 16); SRW(q) = (uint16_t)r[u].y; }
                    }
                }
            } else
            for (uint32_t q0 = 0; q0 < nseg; q0 += kLanes * kBatch) {
                int64_t  sb[kBatch];
                uint32_t sl[kBatch];
                uint16_t sr[kBatch];
#pragma unroll
                for (uint32_t u = 0; u < kBatch; u++) {
                    const uint32_t q = q0 + u * kLanes + lane;
                    sb[u] = q < nseg ? seg_begin[sbase + q] : -1;
                    sl[u] = q < nseg ? seg_len[sbase + q] : 0u;
                    sr[u] = q < nseg ? seg_row[sbase + q] : (uint16_t)0;
                }
#pragma unroll
                for (uint32_t u = 0; u < kBatch; u++) {
                    const uint32_t q = q0 + u * kLanes + lane;
                    if (q < nseg) { SBEG(q) = sb[u] < 0 ? (uint16_t)0xffffu : (uint16_t)(sb[u] - b); SLEN(q) = (uint16_t)sl[u]; SRW(q) = sr[u]; }
                }
            }
        } else {
            for (uint32_t i0 = 0; i0 <= nrow_seg; i0 += kLanes * kBatch) {
                int64_t a[kBatch];
#pragma unroll
                for (uint32_t u = 0; u < kBatch; u++) {
                    const uint32_t i = i0 + u * kLanes + lane;
                    a[u] = i <= nrow_seg ? rp[(int64_t)row_first + i] : 0;
                }
#pragma unroll
                for (uint32_t u = 0; u < kBatch; u++) {
                    const uint32_t i = i0 + u * kLanes + lane;
                    if (i <= nrow_seg) { int64_t x = a[u] > b ? a[u] : b; x = x < e ? x : e; SROW(i) = (uint32_t)(x - b); }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the wavefront's own LDS writes, in order, before its reads below
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    CONV_CLOCK(1);
    for (int g = 0; g < G; g++) {
        uint32_t cw[4], cj[4], tg4[4];
        T        vv[4];
        int64_t  pj[4];
#pragma unroll
        for (int j = 0; j < kGroupSteps; j++) {
            const uint64_t em = __ballot(cnt == 0);
            if (em) {
                const uint32_t rank = lane_rank(em);
                const uint32_t nempty = (uint32_t)__popcll(em);
                const uint32_t navail = nseg - fed;
                bool           want = cnt == 0;
                if (want && rank < navail) {                 // feeding, spmv.cpp:821-868
                    const uint32_t q = fed + rank;
                    if constexpr (SEGT && STAGE) {
                        const uint32_t sb = SBEG(q);
                        pos = sb == 0xffffu ? -1 : b + (int64_t)sb;
                        cnt = SLEN(q);
                        rowtag = TAG ? (uint32_t)SRW(q) : (uint32_t)SRW(q) << col_bits;
                    } else if constexpr (SEGT) {
                        if (seg_packed) {
                            const uint2 r = reinterpret_cast<const uint2 *>(seg_begin)[sbase + q];
                            pos = (r.x & 0xffffu) == 0xffffu ? -1 : b + (int64_t)(r.x & 0xffffu);
                            cnt = r.x >> 16;
                            rowtag = TAG ? r.y : r.y << col_bits;
                        } else {
                            pos = seg_begin[sbase + q];
                            cnt = seg_len[sbase + q];
                            rowtag = TAG ? (uint32_t)seg_row[sbase + q] : (uint32_t)seg_row[sbase + q] << col_bits;
                        }
                    } else if constexpr (STAGE) {
                        if (q < nrow_seg) {
                            const uint32_t a = SROW(q), z = SROW(q + 1);
                            if (z > a) { pos = b + (int64_t)a; cnt = z - a; }
                            else { pos = -1; cnt = 1; }          // empty row: one pad slot
                        } else { pos = -1; cnt = pc; }           // the chunk's trailing pad segment
                    } else if (q < nrow_seg) {
                        const int64_t r = (int64_t)row_first + q;
                        int64_t       a = rp[r], z = rp[r + 1];
                        a = a > b ? a : b;                   // a row begun in an earlier chunk (spmv.cpp:748-756)
                        z = z < e ? z : e;                   // a row continued in the next one (spmv.cpp:851)
                        if (z > a) { pos = a; cnt = (uint32_t)(z - a); }
                        else { pos = -1; cnt = 1; }          // empty row: one pad slot
                    } else { pos = -1; cnt = pc; }           // the chunk's trailing pad segment
                    want = false;
                }
                fed += nempty < navail ? nempty : navail;
                if (nempty > navail) {                       // stealing, spmv.cpp:869-943
                    const uint32_t ave = (uint32_t)(S - (g * kGroupSteps + j));   // == sum(count)/64, App. A.6
                    const uint32_t nsteal = nempty - navail;
                    const uint32_t s = rank - navail;        // steal order = lane order (spmv.cpp:814)
                    uint32_t       base = 0;
                    while (base < nsteal) {
                        const uint64_t of = __ballot(cnt > ave);
                        if (!of) { bad |= 2u; break; }
                        const int      v = __ffsll((unsigned long long)of) - 1;     // FIRST over-full lane (spmv.cpp:876-879)
                        const uint32_t cv = __shfl(cnt, v);
                        const int64_t  pv = __shfl(pos, v);
                        uint32_t       tv = 0;                        // column phases: what a lane steals keeps the row of the victim's segment
                        if constexpr (SEGT) tv = __shfl(rowtag, v);
                        const uint32_t m = (cv + ave - 1) / ave - 1;  // steals until v is no longer over-full
                        const uint32_t kk = m < nsteal - base ? m : nsteal - base;
                        if (want && s >= base && s < base + kk) {     // takes the FIRST ave (spmv.cpp:927-931)
                            pos = pv < 0 ? -1 : pv + (int64_t)(s - base) * ave;
                            cnt = ave;
                            rowtag = tv;
                            tgt = (uint32_t)v;
                            want = false;
                        }
                        if (lane == (uint32_t)v) {
                            if (pos >= 0) pos += (int64_t)kk * ave;
                            cnt -= kk * ave;
                        }
                        base += kk;
                    }
                }
            }
            // which element the lane emits at this step; the element itself is fetched below, for the four steps together: the
            // hand-out of the next steps depends on the counts only, not on what was loaded
            pj[j] = pos;
            if constexpr (TAG) { cw[j] = cnt == 1 ? kEndBit : 0u; tg4[j] = cnt == 1 ? rowtag : 0u; }
            else cw[j] = cnt == 1 ? kEndBit | rowtag : 0u;
            if (pos >= 0) pos++;
            cnt--;
        }
        if (g == 3) CONV_CLOCK(2);
#pragma unroll
        for (int j = 0; j < kGroupSteps; j++) {
            cj[j] = pj[j] >= 0 ? (uint32_t)cidx[pj[j]] : pad_col;
            vv[j] = pj[j] >= 0 ? vals[pj[j]] : (T)0;
        }
#pragma unroll
        for (int j = 0; j < kGroupSteps; j++) {
            uint32_t c = cj[j];
            if (hub_index && pj[j] >= 0 && ((hub_bitmap[c >> 5] >> (c & 31u)) & 1u)) {      // hub column: its index in the LDS table (full order: every column's rank)
                const uint32_t rk = (uint32_t)hub_index[c];
                c = rk < hub_n ? kHubBit | rk : rk;
            }
            cw[j] |= c;
        }
        uint8_t *o = out + (size_t)g * GB;
        if constexpr (C16) {            // 16-bit offsets from the chunk's smallest column; 0x7fff = the pad column; bit 15 = end of segment
            uint32_t h[4];
#pragma unroll
            for (int j = 0; j < kGroupSteps; j++) {
                const uint32_t c = cw[j] & kColMask;
                h[j] = (c == pad_col ? kC16Pad : c - base_col) | ((cw[j] & kEndBit) ? 0x8000u : 0u);
            }
            uint2 cq = {h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
            *reinterpret_cast<uint2 *>(o) = cq;
            o -= lane * 8;              // values are addressed from the group's start + lane * 16 below
            o += lane * 16;
        } else {
            u32x4 cq = {cw[0], cw[1], cw[2], cw[3]};
            *reinterpret_cast<u32x4 *>(o) = cq;
            if constexpr (TAG) {
                const uint2 tq = {tg4[0] | (tg4[1] << 16), tg4[2] | (tg4[3] << 16)};
                *reinterpret_cast<uint2 *>(stream + (size_t)k * G * GB + (size_t)g * GB + kColsBytes + lane * 8) = tq;
            }
        }
        if constexpr (DICT) {
            uint32_t codes = 0;
#pragma unroll
            for (int j = 0; j < kGroupSteps; j++) {
                const uint32_t cd = dict_code<T>(dict, ndict, vv[j]);
                if (cd >= ndict || dict[cd] != __builtin_bit_cast(bits_t, vv[j])) bad |= 4u;   // value missing from the dictionary
                codes |= (cd & 0xffu) << (8 * j);
            }
            *reinterpret_cast<uint32_t *>(stream + (size_t)k * G * GB + (size_t)g * GB + CB + lane * 4) = codes;
        } else if constexpr (sizeof(T) == 8) {
            f64x2 lo = {vv[0], vv[1]}, hi = {vv[2], vv[3]};
            *reinterpret_cast<f64x2 *>(o + CB) = lo;
            *reinterpret_cast<f64x2 *>(o + CB + kLanes * 16) = hi;
        } else {
            f32x4 vq = {vv[0], vv[1], vv[2], vv[3]};
            *reinterpret_cast<f32x4 *>(o + CB) = vq;
        }
        if (g == 3) CONV_CLOCK(3);
    }
    CONV_CLOCK(4);
    if (cnt != 0) bad |= 1u;
    if (bad) atomicOr(err, bad);
    target[(size_t)k * kLanes + lane] = (uint8_t)tgt;
#undef SBEG
#undef SLEN
#undef SRW
#undef SROW
}


// ---- conversion from LDS (packed segment tables) ----------------------------------------------------------------------
// convert_kernel fetches what a lane emits -- column index and value of ITS position in the CSR arrays -- with per-lane loads: 64
// addresses in up to 64 cache lines per instruction, twice per step.  The chunk's 12 KB of columns stay in L1, but the L1 looks one
// line up per clock: 10.8 M lane requests on 256 L1s are 18 us at best, 40-50 us as measured (CVR_CONVERT_CLOCKS: 8-13 us per 16
// steps of a chunk).  Here the wavefront first copies its chunk's columns and values into LDS with coalesced loads -- the values as
// dictionary codes already, one byte each -- and the per-lane accesses become ds_read.  The feed table passes through a ring of
// 2 x kRingHalf entries in LDS: the half behind the one in use is fetched into registers while that one is consumed (a table in LDS
// as a whole would leave four chunks per CU where seven want to run).  Same hand-out, same image.
constexpr uint32_t kRingHalf = 448, kRingLoads = kRingHalf / kLanes;

template <typename T, bool DICT, bool TAG>
__global__ __launch_bounds__(kLanes) void convert_lds_kernel(
    const int32_t *__restrict__ cidx, const T *__restrict__ vals, const int64_t *__restrict__ nzb,
    const uint4 *__restrict__ desc, const uint2 *__restrict__ desc2, uint8_t *__restrict__ stream, uint8_t *__restrict__ target, uint32_t *__restrict__ err,
    int G, uint32_t nchunks, const uint32_t *__restrict__ nchunks_dev, uint32_t pad_col, const T *__restrict__ dict_g, uint32_t ndict,
    const uint2 *__restrict__ seg_packed, uint32_t col_bits, const uint32_t *__restrict__ seg_flags, const uint8_t *__restrict__ codes_g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t csm[];      // ring uint2 [2 kRingHalf] | dictionary [ndict -> 4] | columns u32 [64 S] | values T [64 S] or codes u8 [64 S]
    typedef typename Bits<T>::type bits_t;
    if (seg_flags[0] & 3u) return;
    if (nchunks_dev) nchunks = min(nchunks, *nchunks_dev);      // (a plan beyond the launch's room: the host falls back)
    constexpr int GB = (DICT ? kGroupBytesDict : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    constexpr int CB = kColsBytes + (TAG ? kTagBytes : 0);
    const uint32_t lane = threadIdx.x, k = blockIdx.x;
    if (k >= nchunks) return;
    CONV_CLOCK(0);
    const int      S = G * kGroupSteps;
    const uint32_t cap = (uint32_t)S * kLanes, dpad = DICT ? (ndict + 3u) & ~3u : 0u;
    uint2    *ring = reinterpret_cast<uint2 *>(csm);
    bits_t   *dict = reinterpret_cast<bits_t *>(csm + 16u * kRingHalf);
    uint32_t *lcol = reinterpret_cast<uint32_t *>(csm + 16u * kRingHalf + sizeof(bits_t) * dpad);
    uint8_t  *lcode = reinterpret_cast<uint8_t *>(lcol + cap);
    T        *lval = reinterpret_cast<T *>(lcol + cap);
    if constexpr (DICT) {
        for (uint32_t i = lane; i < ndict; i += kLanes) dict[i] = __builtin_bit_cast(bits_t, dict_g[i]);
    }
    const int64_t  b = nzb[k], e = nzb[k + 1];
    const uint32_t n = (uint32_t)(e - b), nseg = desc[k].y, sbase = desc2[k].x;
    const uint2   *pk = seg_packed + sbase;
    // the ring's two halves and the half behind them (registers), then the chunk's columns and values, eight loads in flight per lane
    uint2 pend[kRingLoads];
#pragma unroll
    for (uint32_t u = 0; u < 2 * kRingLoads; u++) { const uint32_t i = u * kLanes + lane; ring[i] = i < nseg ? pk[i] : uint2{0u, 0u}; }
#pragma unroll
    for (uint32_t u = 0; u < kRingLoads; u++) { const uint32_t i = 2 * kRingHalf + u * kLanes + lane; pend[u] = i < nseg ? pk[i] : uint2{0u, 0u}; }
    uint32_t ring_end = 2 * kRingHalf, bad = 0;        // entries [ring_end - 2 kRingHalf, ring_end) are in the ring, pend holds the kRingHalf behind them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the dictionary, before its use below)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // columns and values, four consecutive elements per lane and load (16 bytes of columns, 16 or 32 of values); kVec such loads of
    // each are issued before the first is used; the dictionary searches of a batch run side by side
    constexpr uint32_t kVec = 4;
    if (DICT && codes_g) {
        // the values come as dictionary codes already (dict_codes_kernel: one coalesced pass over the values beside the planner): a byte
        // per element instead of eight, and no search here
        for (uint32_t i0 = lane * 4; i0 < n; i0 += kLanes * 4 * kVec) {
            u32x4    c[kVec];
            uint32_t cd[kVec];
#pragma unroll
            for (uint32_t u = 0; u < kVec; u++) {
                const uint32_t i = i0 + u * kLanes * 4;
                cd[u] = 0;
                if (i + 4 <= n) {
                    c[u] = *reinterpret_cast<const u32x4_dw *>(cidx + b + i);
#pragma unroll
                    for (int q = 0; q < 4; q++) cd[u] |= (uint32_t)codes_g[b + i + q] << (8 * q);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) { c[u][q] = i + q < n ? (uint32_t)cidx[b + i + q] : 0u; cd[u] |= (i + q < n ? (uint32_t)codes_g[b + i + q] : 0u) << (8 * q); }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < kVec; u++) {
                const uint32_t i = i0 + u * kLanes * 4;
                if (i >= n) break;
                *reinterpret_cast<u32x4 *>(lcol + i) = c[u];
                *reinterpret_cast<uint32_t *>(lcode + i) = cd[u];
            }
        }
    } else
    for (uint32_t i0 = lane * 4; i0 < n; i0 += kLanes * 4 * kVec) {
        u32x4 c[kVec];
        T     v[kVec][4];
#pragma unroll
        for (uint32_t u = 0; u < kVec; u++) {
            const uint32_t i = i0 + u * kLanes * 4;
            if (i + 4 <= n) {
                c[u] = *reinterpret_cast<const u32x4_dw *>(cidx + b + i);
#pragma unroll
                for (int q = 0; q < 4; q++) v[u][q] = vals[b + i + q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) { c[u][q] = i + q < n ? (uint32_t)cidx[b + i + q] : 0u; v[u][q] = i + q < n ? vals[b + i + q] : (T)0; }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kVec; u++) {
            const uint32_t i = i0 + u * kLanes * 4;
            if (i >= n) break;
            *reinterpret_cast<u32x4 *>(lcol + i) = c[u];          // (room for 64 S elements: the last vector of a short chunk stays inside)
        }
        if constexpr (DICT) {
            // first entry >= the value's bit pattern, for the batch's 4 kVec values at once (the search is a chain of dependent LDS reads)
            uint32_t lo[kVec][4], hi[kVec][4];
#pragma unroll
            for (uint32_t u = 0; u < kVec; u++)
#pragma unroll
                for (int q = 0; q < 4; q++) { lo[u][q] = 0; hi[u][q] = ndict; }
            for (uint32_t span = ndict; span > 0; span >>= 1) {
#pragma unroll
                for (uint32_t u = 0; u < kVec; u++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (lo[u][q] < hi[u][q]) {
                            const uint32_t mid = (lo[u][q] + hi[u][q]) >> 1;
                            if (dict[mid] < __builtin_bit_cast(bits_t, v[u][q])) lo[u][q] = mid + 1; else hi[u][q] = mid;
                        }
                    }
            }
#pragma unroll
            for (uint32_t u = 0; u < kVec; u++) {
                const uint32_t i = i0 + u * kLanes * 4;
                if (i >= n) break;
                uint32_t codes = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t cd = lo[u][q];
                    if (i + q < n && (cd >= ndict || dict[cd] != __builtin_bit_cast(bits_t, v[u][q]))) bad |= 4u;   // value missing from the dictionary
                    codes |= (cd & 0xffu) << (8 * q);
                }
                *reinterpret_cast<uint32_t *>(lcode + i) = codes;
            }
        } else {
#pragma unroll
            for (uint32_t u = 0; u < kVec; u++) {
                const uint32_t i = i0 + u * kLanes * 4;
                if (i >= n) break;
#pragma unroll
                for (int q = 0; q < 4; q++) lval[i + q] = v[u][q];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the wavefront's own LDS writes, in order, before its reads below
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    CONV_CLOCK(1);

    uint32_t pos = 0xffffffffu;      // element of the chunk this lane emits next; 0xffffffff = pad slot
    uint32_t cnt = 0, fed = 0, tgt = lane, rowtag = 0;
    uint8_t *out = stream + (size_t)k * G * GB + lane * 16;
    for (int g = 0; g < G; g++) {
        uint32_t cw[4], tg4[4], pj[4];
#pragma unroll
        for (int j = 0; j < kGroupSteps; j++) {
            const uint64_t em = __ballot(cnt == 0);
            if (em) {
                if (fed + kLanes > ring_end && ring_end < nseg) {      // this step may reach past the ring: the half fetched earlier goes in (over entries long handed out), the next is requested
#pragma unroll
                    for (uint32_t u = 0; u < kRingLoads; u++) ring[(ring_end + u * kLanes + lane) % (2 * kRingHalf)] = pend[u];
                    ring_end += kRingHalf;
#pragma unroll
                    for (uint32_t u = 0; u < kRingLoads; u++) { const uint32_t i = ring_end + u * kLanes + lane; pend[u] = i < nseg ? pk[i] : uint2{0u, 0u}; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                const uint32_t rank = lane_rank(em);
                const uint32_t nempty = (uint32_t)__popcll(em);
                const uint32_t navail = nseg - fed;
                bool           want = cnt == 0;
                if (want && rank < navail) {                 // feeding, spmv.cpp:821-868
                    const uint2 r = ring[(fed + rank) % (2 * kRingHalf)];
                    pos = (r.x & 0xffffu) == 0xffffu ? 0xffffffffu : (r.x & 0xffffu);
                    cnt = r.x >> 16;
                    rowtag = TAG ? r.y : r.y << col_bits;
                    want = false;
                }
                fed += nempty < navail ? nempty : navail;
                if (nempty > navail) {                       // stealing, spmv.cpp:869-943
                    const uint32_t ave = (uint32_t)(S - (g * kGroupSteps + j));   // == sum(count)/64, App. A.6
                    const uint32_t nsteal = nempty - navail;
                    const uint32_t s = rank - navail;        // steal order = lane order (spmv.cpp:814)
                    uint32_t       base = 0;
                    while (base < nsteal) {
                        const uint64_t of = __ballot(cnt > ave);
                        if (!of) { bad |= 2u; break; }
                        const int      v = __ffsll((unsigned long long)of) - 1;     // FIRST over-full lane (spmv.cpp:876-879)
                        const uint32_t cv = __shfl(cnt, v), pv = __shfl(pos, v), tv = __shfl(rowtag, v);
                        const uint32_t m = (cv + ave - 1) / ave - 1;  // steals until v is no longer over-full
                        const uint32_t kk = m < nsteal - base ? m : nsteal - base;
                        if (want && s >= base && s < base + kk) {     // takes the FIRST ave (spmv.cpp:927-931)
                            pos = pv == 0xffffffffu ? pv : pv + (s - base) * ave;
                            cnt = ave;
                            rowtag = tv;
                            tgt = (uint32_t)v;
                            want = false;
                        }
                        if (lane == (uint32_t)v) {
                            if (pos != 0xffffffffu) pos += kk * ave;
                            cnt -= kk * ave;
                        }
                        base += kk;
                    }
                }
            }
            pj[j] = pos;
            if constexpr (TAG) { cw[j] = cnt == 1 ? kEndBit : 0u; tg4[j] = cnt == 1 ? rowtag : 0u; }
            else cw[j] = cnt == 1 ? kEndBit | rowtag : 0u;
            if (pos != 0xffffffffu) pos++;
            cnt--;
        }
        if (g == 3) CONV_CLOCK(2);
        uint8_t *o = out + (size_t)g * GB;
#pragma unroll
        for (int j = 0; j < kGroupSteps; j++) cw[j] |= pj[j] != 0xffffffffu ? lcol[pj[j]] : pad_col;
        u32x4 cq = {cw[0], cw[1], cw[2], cw[3]};
        *reinterpret_cast<u32x4 *>(o) = cq;
        if constexpr (TAG) {
            const uint2 tq = {tg4[0] | (tg4[1] << 16), tg4[2] | (tg4[3] << 16)};
            *reinterpret_cast<uint2 *>(stream + (size_t)k * G * GB + (size_t)g * GB + kColsBytes + lane * 8) = tq;
        }
        if constexpr (DICT) {
            uint32_t codes = 0;             // (a pad slot holds +0.0: the dictionary's first entry)
#pragma unroll
            for (int j = 0; j < kGroupSteps; j++) codes |= (pj[j] != 0xffffffffu ? (uint32_t)lcode[pj[j]] : 0u) << (8 * j);
            *reinterpret_cast<uint32_t *>(stream + (size_t)k * G * GB + (size_t)g * GB + CB + lane * 4) = codes;
        } else {
            T vv[4];
#pragma unroll
            for (int j = 0; j < kGroupSteps; j++) vv[j] = pj[j] != 0xffffffffu ? lval[pj[j]] : (T)0;
            if constexpr (sizeof(T) == 8) {
                f64x2 lo = {vv[0], vv[1]}, hi = {vv[2], vv[3]};
                *reinterpret_cast<f64x2 *>(o + CB) = lo;
                *reinterpret_cast<f64x2 *>(o + CB + kLanes * 16) = hi;
            } else {
                f32x4 vq = {vv[0], vv[1], vv[2], vv[3]};
                *reinterpret_cast<f32x4 *>(o + CB) = vq;
            }
        }
        if (g == 3) CONV_CLOCK(3);
    }
    CONV_CLOCK(4);
    if (cnt != 0) bad |= 1u;
    if (bad) atomicOr(err, bad);
    target[(size_t)k * kLanes + lane] = (uint8_t)tgt;
}

// code[j] = the dictionary code of vals[j] (j in [n0, n1)): one coalesced pass, sixteen values per thread and trip, the searches of a
// trip side by side; a value that is not in the dictionary sets *err bit 2 (the converter's flag)
template <typename T>
__global__ __launch_bounds__(256) void dict_codes_kernel(const T *__restrict__ vals, long long n0, long long n1, const T *__restrict__ dict_g, uint32_t ndict,
                                                         uint8_t *__restrict__ codes, uint32_t *__restrict__ err)
{
    typedef typename Bits<T>::type bits_t;
    __shared__ bits_t dict[kDictMax];
    for (uint32_t i = threadIdx.x; i < ndict; i += blockDim.x) dict[i] = __builtin_bit_cast(bits_t, dict_g[i]);
    __syncthreads();
    // four consecutive values per thread and load (a wavefront reads 2 KiB in a row), four such loads in flight, their sixteen searches side by side
    constexpr int kVec = 4;
    uint32_t      bad = 0;
    const long long tstride = (long long)gridDim.x * blockDim.x * 4;
    for (long long j0 = n0 + ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; j0 < n1; j0 += tstride * kVec) {
        bits_t   v[kVec][4];
        uint32_t lo[kVec][4], hi[kVec][4];
#pragma unroll
        for (int u = 0; u < kVec; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const long long j = j0 + u * tstride + q;
                v[u][q] = j < n1 ? __builtin_bit_cast(bits_t, vals[j]) : (bits_t)0; lo[u][q] = 0; hi[u][q] = ndict;
            }
        for (uint32_t span = ndict; span > 0; span >>= 1) {
#pragma unroll
            for (int u = 0; u < kVec; u++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (lo[u][q] < hi[u][q]) { const uint32_t mid = (lo[u][q] + hi[u][q]) >> 1; if (dict[mid] < v[u][q]) lo[u][q] = mid + 1; else hi[u][q] = mid; }
        }
#pragma unroll
        for (int u = 0; u < kVec; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const long long j = j0 + u * tstride + q;
                if (j < n1) {
                    if (lo[u][q] >= ndict || dict[lo[u][q]] != v[u][q]) bad = 4u;
                    codes[j] = (uint8_t)lo[u][q];
                }
            }
    }
    if (bad) atomicOr(err, bad);
}

// ---- column phases: the segment table -------------------------------------------------------------------------------
// A chunk with column phases feeds, phase by phase (column range by column range), the pieces of its rows that fall into
// the phase: one segment per (row, phase) pair with a non-zero, in (phase, row) order; an empty row keeps its pad slot
// (phase 0); the chunk's trailing pad segment comes last.  Rows must have ascending columns (flags[0] bit 0 otherwise).

// the piece of row i of chunk k inside the chunk's CSR range [b, e)
__device__ __forceinline__ void row_piece(const int64_t *rp, uint32_t row, int64_t b, int64_t e, int64_t &a, int64_t &z)
{
    a = rp[row]; z = rp[row + 1];
    a = a > b ? a : b;
    z = z < e ? z : e;
}

// Both passes stage the chunk's column indices (as many of the chunk's elements as the launch reserved room for: lds_cols) and its row pieces in LDS:
// the per-row scans and binary searches then run at LDS latency instead of as chains of dependent global loads.
// (the launch reserves 5 bytes per element of the chunk: column index + row-start flag, for as many elements as the LDS holds beside the row pieces: `lds_cols`)
constexpr uint32_t kSegWaves = 8;           // wavefronts per chunk in the fill pass (each takes every 8th phase)

struct SegStage {
    const int32_t *cols;     // the chunk's column indices, indexed from the chunk's first CSR element
    uint32_t      *pa, *pz;  // per row of the chunk: its piece [pa, pz) relative to the chunk's first element (pa == pz: empty row)
};

__device__ __forceinline__ SegStage seg_stage(uint8_t *smem, const int64_t *rp, const int32_t *cidx, int64_t b, int64_t e,
                                              uint32_t row_first, uint32_t nri, uint32_t lds_cols)
{
    SegStage st;
    st.pa = reinterpret_cast<uint32_t *>(smem);
    st.pz = st.pa + nri;
    int32_t       *scol = reinterpret_cast<int32_t *>(st.pz + nri);
    const uint32_t n = (uint32_t)(e - b);
    for (uint32_t i = threadIdx.x; i < nri; i += blockDim.x) {
        int64_t a, z;
        row_piece(rp, row_first + i, b, e, a, z);
        if (z < a) z = a;
        st.pa[i] = (uint32_t)(a - b); st.pz[i] = (uint32_t)(z - b);
    }
    if (n <= lds_cols) {
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) scol[j] = cidx[b + j];
        st.cols = scol;
    } else {
        st.cols = cidx + b;
    }
    __syncthreads();
    return st;
}

// col / pw without the integer division (a dozen instructions per element of a kernel that is all such small steps): the float
// quotient, corrected by at most one either way (col < 2^31, pw >= 16: the float error stays below one unit of the quotient)
__device__ __forceinline__ uint32_t phase_of(uint32_t col, uint32_t pw, float inv_pw)
{
    uint32_t q = (uint32_t)((float)col * inv_pw);
    if ((uint64_t)q * pw > col) q--;
    else if ((uint64_t)(q + 1) * pw <= col) q++;
    return q;
}

// first position in cols[a, z) whose column is >= bound (columns ascending)
__device__ __forceinline__ uint32_t lower_col(const int32_t *cols, uint32_t a, uint32_t z, uint64_t bound)
{
    while (a < z) {
        const uint32_t mid = (a + z) >> 1;
        if ((uint64_t)(uint32_t)cols[mid] < bound) a = mid + 1; else z = mid;
    }
    return a;
}

// The segment table of one chunk, one workgroup (8 wavefronts): (1) the number of segments of every phase, one thread per element
// (an element starts a segment if it starts its row's piece or lies in another phase than its predecessor; row starts are
// flagged in LDS behind the columns), turned into offsets inside the chunk; (2) wavefront w writes the segments of phases w,
// w + 8, ...: rows in order, ballot compaction.  The chunk's segments start at k * 64 S of the table (a chunk has at most 64 S),
// so no pass over all chunks is needed between counting and filling, and the chunk's columns and row pieces are staged once.
__global__ __launch_bounds__(kLanes * kSegWaves) void seg_build_kernel(const int64_t *__restrict__ rp, const int32_t *__restrict__ cidx,
                                                          const int64_t *__restrict__ nzb, const uint32_t *__restrict__ pad_cnt,
                                                          uint4 *__restrict__ desc, uint2 *__restrict__ desc2, uint32_t nchunks,
                                                          uint32_t pw, uint32_t phases, uint32_t cap, uint32_t *__restrict__ cnt, int64_t *__restrict__ seg_begin,
                                                          uint32_t *__restrict__ seg_len, uint16_t *__restrict__ seg_row, uint32_t *__restrict__ flags, uint32_t lds_cols, uint32_t psh)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t pc[64], poff[64], sbad, stotal;
    const uint32_t k = blockIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (k >= nchunks) return;
    const float    inv_pw = 1.0f / (float)pw;
    const uint32_t pmask = (1u << psh) - 1u;       // pieces of at most 2^psh elements (psh = 31: whole segments; a chunk has fewer than 2^31 elements)
    const int64_t  b = nzb[k], e = nzb[k + 1];
    const uint32_t row_first = desc[k].x, nri = desc2[k].y, sbase = k * cap;
    if (threadIdx.x < 64) pc[threadIdx.x] = 0;
    if (threadIdx.x == 0) sbad = 0;
    const SegStage st = seg_stage(smem, rp, cidx, b, e, row_first, nri, lds_cols);
    uint32_t       bad = 0;
    const uint32_t n = (uint32_t)(e - b);
    if (n <= lds_cols) {
        uint8_t *rstart = smem + 8 * (size_t)nri + 4 * (size_t)n;
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) rstart[j] = 0;
        __syncthreads();
        uint32_t empty = 0;
        for (uint32_t i = threadIdx.x; i < nri; i += blockDim.x) {
            if (st.pz[i] > st.pa[i]) rstart[st.pa[i]] = 1; else empty++;          // an empty row owns one pad slot, fed with phase 0
        }
        if (empty) atomicAdd(&pc[0], empty);
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
            const int32_t  col = st.cols[j];
            const uint32_t ph = phase_of((uint32_t)col, pw, inv_pw);
            if (rstart[j]) { atomicAdd(&pc[ph], 1u); continue; }
            const int32_t prev_col = st.cols[j - 1];      // (j > 0 here: element 0 starts a row piece or belongs to no row of the chunk)
            if (col < prev_col) bad = 1;
            if (ph != phase_of((uint32_t)prev_col, pw, inv_pw) || (j & pmask) == 0) atomicAdd(&pc[ph], 1u);      // (pieces are cut at the multiples of 2^psh from the chunk's first element)
        }
    } else {
        for (uint32_t i = threadIdx.x; i < nri; i += blockDim.x) {
            const uint32_t a = st.pa[i], z = st.pz[i];
            if (z <= a) { atomicAdd(&pc[0], 1u); continue; }
            int32_t  prev_col = st.cols[a];
            uint32_t prev = phase_of((uint32_t)prev_col, pw, inv_pw);
            atomicAdd(&pc[prev], 1u);
            for (uint32_t j = a + 1; j < z; j++) {
                const int32_t  col = st.cols[j];
                const uint32_t ph = phase_of((uint32_t)col, pw, inv_pw);
                if (col < prev_col) bad = 1;
                if (ph != prev || (j & pmask) == 0) atomicAdd(&pc[ph], 1u);
                prev = ph; prev_col = col;
            }
        }
    }
    if (bad) sbad = 1;
    __syncthreads();
    const uint32_t padc = pad_cnt[k];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t p = 0; p < phases; p++) { const uint32_t c = pc[p]; poff[p] = run; run += c; }
        run += padc > 0 ? 1u : 0u;
        stotal = run;
        cnt[k] = run;
        desc[k].y = run;
        desc2[k].x = sbase;
        if (sbad) atomicOr(&flags[0], 1u);      // (rare; no other same-address atomic here: thousands of workgroups on one word cost ~100 ns each)
    }
    __syncthreads();
    if (sbad) return;                            // unsorted rows: the table is meaningless (cvr_preprocess reports it)
    for (uint32_t p = wv; p < phases; p += kSegWaves) {
        const uint64_t c0 = (uint64_t)p * pw, c1 = p + 1 == phases ? ~0ull : c0 + pw;
        uint32_t       q = sbase + poff[p];
        for (uint32_t r0 = 0; r0 < nri; r0 += kLanes) {
            const uint32_t i = r0 + lane;
            uint32_t       np = 0, lo = 0, hi = 0;              // pieces of row i in this phase: [lo, hi) cut at the multiples of pmax
            bool           pad = false;
            if (i < nri) {
                const uint32_t a = st.pa[i], z = st.pz[i];
                if (z <= a) { pad = p == 0; np = pad ? 1u : 0u; }
                else {
                    lo = lower_col(st.cols, a, z, c0); hi = lower_col(st.cols, lo, z, c1);
                    if (hi > lo) np = ((hi - 1) >> psh) - (lo >> psh) + 1;
                }
            }
            uint32_t incl = np;                                   // inclusive scan over the wavefront
#pragma unroll
            for (int o = 1; o < kLanes; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
            uint32_t idx = q + incl - np;
            if (pad) { seg_begin[idx] = -1; seg_len[idx] = 1; seg_row[idx] = (uint16_t)i; }
            else
                for (uint32_t u = lo; u < hi;) {
                    const uint32_t nxt = ((u >> psh) + 1) << psh, v = nxt < hi ? nxt : hi;
                    seg_begin[idx] = b + u; seg_len[idx] = v - u; seg_row[idx] = (uint16_t)i;
                    idx++; u = v;
                }
            q += __shfl(incl, kLanes - 1);
        }
    }
    if (padc > 0 && threadIdx.x == 0) {
        const uint32_t q = sbase + stotal - 1;
        seg_begin[q] = -1; seg_len[q] = padc; seg_row[q] = (uint16_t)nri;
    }
}

// The same table as a stable counting sort of the chunk's pieces by phase, for chunks whose elements fit the LDS with their bookkeeping
// (seg_scan_lds_bytes).  The search-based kernel above and a first version of this one (windows of 64 elements, one ballot per phase
// for the ranks) were bound by instruction issue: 500 wavefront instructions per 64 elements = 42 M for the web-Google shape, 73 us
// on 1 024 SIMDs that issue one wave64 instruction per four clocks.  Here every thread owns a run of L consecutive elements:
//   rows     pieces [pa, pz) of the chunk's rows; E = empty rows in front of a row (an empty row owns a pad slot, fed in phase 0)
//   walk 1   an element starts a piece if it starts its row (marks scattered by the rows), lies in another phase than its predecessor
//            or sits at a multiple of the piece length; the thread counts its starts per phase in a column of its own of hist[phase][thread]
//   scans    hist in (phase, thread) order -> where a thread's pieces of a phase go; the last row start and the first piece start
//            of the runs in front of / behind a run (running maximum / minimum over the threads)
//   walk 2   a piece's place = hist[phase][thread]++ (+ the empty rows in front of its row in phase 0, all of them in the later
//            phases: the table is in (phase, row, position) order); its length = distance to the next start
constexpr uint32_t kSegThreads = 256, kSegRunMax = 32;

__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int o = 1; o < kLanes; o <<= 1) { const uint32_t t = __shfl_up(v, o); if ((int)lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int o = 1; o < kLanes; o <<= 1) { const uint32_t t = __shfl_up(v, o); if ((int)lane >= o) v = v > t ? v : t; }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_min_down(uint32_t v, uint32_t lane)      // minimum over this lane and the lanes behind it
{
#pragma unroll
    for (int o = 1; o < kLanes; o <<= 1) { const uint32_t t = __shfl_down(v, o); if ((int)lane + o < kLanes) v = v < t ? v : t; }
    return v;
}

__global__ __launch_bounds__(kSegThreads) void seg_scan_kernel(const int64_t *__restrict__ rp, const int32_t *__restrict__ cidx,
                                                         const int64_t *__restrict__ nzb, const uint32_t *__restrict__ pad_cnt,
                                                         uint4 *__restrict__ desc, uint2 *__restrict__ desc2, uint32_t nchunks, const uint32_t *__restrict__ nchunks_dev,
                                                         uint32_t pw, uint32_t phases, uint32_t cap, uint32_t *__restrict__ cnt, uint2 *__restrict__ packed,
                                                         uint32_t *__restrict__ flags, uint32_t psh, unsigned long long *__restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t kWaves = kSegThreads / kLanes;
    __shared__ uint32_t sbad, etotal, wsum[kWaves], wmax[kWaves], wmin[kWaves];
#define SEG_CLOCK(i) do { if (dbg && threadIdx.x == 0 && (blockIdx.x & 255u) == 0) dbg[(blockIdx.x >> 8) * 16 + (i)] = wall_clock64(); } while (0)
    SEG_CLOCK(0);
    const uint32_t k = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (k >= (nchunks_dev ? min(nchunks, *nchunks_dev) : nchunks)) return;
    const float    inv_pw = 1.0f / (float)pw;
    const uint32_t pmask = (1u << psh) - 1u;
    const int64_t  b = nzb[k], e = nzb[k + 1];
    const uint32_t row_first = desc[k].x, nri = desc2[k].y, sbase = k * cap, n = (uint32_t)(e - b);
    const uint32_t RW = nri / 64 + 1, L = (n + kSegThreads - 1) / kSegThreads;        // row windows; elements per thread
    if (nri > cap || n > cap) { if (tid == 0) atomicOr(&flags[0], 2u); return; }      // (a plan that does not belong to this launch)
    // LDS: epre | cols | hist | E | mark | phb | z0  (sized by the chunk's slots: a chunk of 64 S slots has at most 64 S rows and elements)
    uint32_t *epre = reinterpret_cast<uint32_t *>(smem);                   // [RW] empty rows in front of a window of 64 rows
    int32_t  *cols = reinterpret_cast<int32_t *>(epre + RW);               // [n]
    uint16_t *hist = reinterpret_cast<uint16_t *>(cols + n);               // [phases][kSegThreads]
    uint16_t *E = hist + (size_t)phases * kSegThreads;                     // [nri] empty rows in front of the row inside its window
    uint16_t *mark = E + nri;                                              // [n] row + 1 at the first element of a row, else 0
    uint8_t  *phb = reinterpret_cast<uint8_t *>(mark + n);                 // [n] phase
    uint8_t  *z0 = phb + n;                                                // [n] phase-0 starts of the run in front of the element
    if (tid == 0) { sbad = 0; etotal = 0; }
    // All loads of a batch are issued before the first is used: one round trip per eight windows of rows / 2 048 columns
    constexpr uint32_t kBatch = 8;
    int32_t c0[kBatch];
    int64_t ra[kBatch], rz[kBatch];
#pragma unroll
    for (uint32_t t = 0; t < kBatch; t++) { const uint32_t j = tid + t * kSegThreads; c0[t] = j < n ? cidx[b + j] : 0; }
#pragma unroll
    for (uint32_t t = 0; t < kBatch; t++) {
        const uint32_t i = (wv + t * kWaves) * 64 + lane;
        ra[t] = i < nri ? rp[row_first + i] : 0;
        rz[t] = i < nri ? rp[row_first + i + 1] : 0;
    }
    for (uint32_t j = tid; j < n; j += kSegThreads) mark[j] = 0;
    for (uint32_t p = 0; p < phases; p++) hist[p * kSegThreads + tid] = 0;
    __syncthreads();
    // rows: the first element of a row is marked with the row; empty rows are counted (inside the window of 64 rows; the windows' sums follow)
    for (uint32_t u0 = wv; u0 < RW; u0 += kWaves * kBatch) {
        if (u0 != wv) {
#pragma unroll
            for (uint32_t t = 0; t < kBatch; t++) {
                const uint32_t i = (u0 + t * kWaves) * 64 + lane;
                ra[t] = i < nri ? rp[row_first + i] : 0;
                rz[t] = i < nri ? rp[row_first + i + 1] : 0;
            }
        }
#pragma unroll
        for (uint32_t t = 0; t < kBatch; t++) {
            const uint32_t u = u0 + t * kWaves, i = u * 64 + lane;
            if (u >= RW) break;
            bool empty = false;
            if (i < nri) {
                const int64_t a = ra[t] > b ? ra[t] : b, z = rz[t] < e ? rz[t] : e;
                empty = z <= a;
                if (!empty) mark[(uint32_t)(a - b)] = (uint16_t)(i + 1);
            }
            const unsigned long long m = __ballot(empty);
            if (i < nri) E[i] = (uint16_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) epre[u] = (uint32_t)__popcll(m);
        }
    }
#pragma unroll
    for (uint32_t t = 0; t < kBatch; t++) { const uint32_t j = tid + t * kSegThreads; if (j < n) cols[j] = c0[t]; }
    for (uint32_t j0 = tid + kSegThreads * kBatch; j0 < n; j0 += kSegThreads * kBatch) {
        int32_t c[kBatch];
#pragma unroll
        for (uint32_t t = 0; t < kBatch; t++) { const uint32_t j = j0 + t * kSegThreads; c[t] = j < n ? cidx[b + j] : 0; }
#pragma unroll
        for (uint32_t t = 0; t < kBatch; t++) { const uint32_t j = j0 + t * kSegThreads; if (j < n) cols[j] = c[t]; }
    }
    __syncthreads();
    SEG_CLOCK(1);
    if (wv == 0) {          // empty rows in front of every row window
        uint32_t carry = 0;
        for (uint32_t u0 = 0; u0 < RW; u0 += 64) {
            const uint32_t u = u0 + lane, c = u < RW ? epre[u] : 0u, incl = wave_incl_sum(c, lane);
            if (u < RW) epre[u] = carry + incl - c;
            carry += __shfl(incl, 63);
        }
        if (lane == 0) etotal = carry;
    }
    SEG_CLOCK(2);
    // walk 1
    const uint32_t j0 = tid * L, j1 = min(n, j0 + L);
    uint32_t       smask = 0, last_mark = 0, bad = 0, z = 0, pph = 0;
    int32_t        pcol = 0;
    if (j0 < j1 && j0 > 0) { pcol = cols[j0 - 1]; pph = phase_of((uint32_t)pcol, pw, inv_pw); }
    for (uint32_t j = j0; j < j1; j++) {
        const int32_t  col = cols[j];
        const uint32_t ph = phase_of((uint32_t)col, pw, inv_pw), m = mark[j];
        bool           start = m != 0;
        if (!start) {                                         // (j > 0 here: element 0 starts a row piece)
            if (col < pcol) bad = 1;
            start = ph != pph || (j & pmask) == 0;
        }
        phb[j] = (uint8_t)ph;
        z0[j] = (uint8_t)z;
        if (start) {
            smask |= 1u << (j - j0);
            hist[ph * kSegThreads + tid] += 1;
            if (ph == 0) z++;
        }
        if (m) last_mark = m;
        pcol = col; pph = ph;
    }
    if (bad) sbad = 1;
    // the last row start in front of the run, the first piece start behind it
    const uint32_t first_start = smask ? j0 + (uint32_t)__builtin_ctz(smask) : n;
    const uint32_t imax = wave_incl_max(last_mark, lane), imin = wave_incl_min_down(first_start, lane);
    if (lane == 63) wmax[wv] = imax;
    if (lane == 0) wmin[wv] = imin;
    __syncthreads();
    SEG_CLOCK(3);
    uint32_t rbefore = __shfl_up(imax, 1), nafter = __shfl_down(imin, 1);
    if (lane == 0) rbefore = 0;
    if (lane == 63) nafter = n;
    for (uint32_t w = 0; w < kWaves; w++) {
        if (w < wv) rbefore = max(rbefore, wmax[w]);
        if (w > wv) nafter = min(nafter, wmin[w]);
    }
    // hist in (phase, thread) order: a thread sums `phases` consecutive entries, the sums are scanned over the workgroup
    uint32_t mine = 0;
    for (uint32_t q = 0; q < phases; q++) mine += hist[tid * phases + q];
    const uint32_t isum = wave_incl_sum(mine, lane);
    if (lane == 63) wsum[wv] = isum;
    __syncthreads();
    uint32_t run = isum - mine;
    for (uint32_t w = 0; w < wv; w++) run += wsum[w];
    for (uint32_t q = 0; q < phases; q++) { const uint32_t c = hist[tid * phases + q]; hist[tid * phases + q] = (uint16_t)run; run += c; }
    const uint32_t padc = pad_cnt[k];
    uint32_t       stotal = 0;
    for (uint32_t w = 0; w < kWaves; w++) stotal += wsum[w];
    stotal += etotal + (padc > 0 ? 1u : 0u);
    if (tid == 0) {
        cnt[k] = stotal;
        desc[k].y = stotal;
        desc2[k].x = sbase;
        if (sbad) atomicOr(&flags[0], 1u);
    }
    __syncthreads();
    SEG_CLOCK(4);
    if (sbad) return;                            // unsorted rows: the table is meaningless (cvr_preprocess reports it)
    // walk 2
    const uint32_t et = etotal;
    uint32_t       rcur = rbefore;
    for (uint32_t j = j0; j < j1; j++) {
        const uint32_t m = mark[j], i = j - j0;
        if (m) rcur = m;
        if ((smask >> i) & 1u) {
            const uint32_t ph = phb[j], r = rcur - 1u;
            const uint32_t slot = hist[ph * kSegThreads + tid];
            hist[ph * kSegThreads + tid] = (uint16_t)(slot + 1);
            const uint32_t rest = i < 31 ? smask >> (i + 1) : 0u;
            const uint32_t nxt = rest ? j + 1 + (uint32_t)__builtin_ctz(rest) : nafter;
            const uint32_t idx = sbase + slot + (ph == 0 ? epre[r >> 6] + E[r] : et);
            packed[idx] = uint2{j | ((nxt - j) << 16), r};
        }
    }
    __syncthreads();
    SEG_CLOCK(5);
    // empty rows: behind the phase-0 pieces in front of their position and the empty rows in front of them
    // (hist has moved on to the END of every thread's pieces of a phase: the pieces of phase 0 in front of position a are those
    // of the threads in front of its run -- the start of that thread's range = the end of its predecessor's -- and z0)
    if (et > 0)
        for (uint32_t i = tid; i < nri; i += kSegThreads) {
            int64_t a, z;
            row_piece(rp, row_first + i, b, e, a, z);
            if (z > a) continue;
            const uint32_t at = (uint32_t)(a - b);
            uint32_t       before;
            if (at >= n) before = hist[(size_t)kSegThreads - 1];
            else { const uint32_t t = at / L; before = (t ? hist[t - 1] : 0u) + z0[at]; }
            packed[sbase + before + epre[i >> 6] + E[i]] = uint2{0xffffu | (1u << 16), i};
        }
    if (padc > 0 && tid == 0) packed[sbase + stotal - 1] = uint2{0xffffu | (padc << 16), nri};
    SEG_CLOCK(6);
#undef SEG_CLOCK
}

// sum of the chunks' segment counts (cvr_info.nsegments), one workgroup
__global__ __launch_bounds__(1024) void seg_total_kernel(uint32_t *__restrict__ cnt, uint32_t nchunks, const uint32_t *__restrict__ nchunks_dev, uint32_t *__restrict__ total_out)
{
    if (nchunks_dev) nchunks = min(nchunks, *nchunks_dev);      // (a plan beyond the launch's room: the host falls back)
    __shared__ uint32_t part[16];
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < nchunks; i += 1024) s += cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < 16; w++) t += part[w]; *total_out = t; }
}

// Window choice for the SpMV kernel's LDS staging of x: one workgroup per SpMV workgroup (wpb
// consecutive chunks = one contiguous CSR range).  Histogram of the range's columns over bins of 2^binshift
// columns in LDS, then the best run of `nb` consecutive bins; ties go to the lowest column.
__global__ __launch_bounds__(1024) void window_kernel(const int32_t *__restrict__ cidx, const int64_t *__restrict__ nzb,
                                                      uint32_t nchunks, uint32_t ncols1, uint32_t wn, uint32_t binshift,
                                                      uint32_t nbins, uint32_t nb, uint32_t *__restrict__ win_base, uint32_t wpb, const uint32_t *__restrict__ nchunks_dev)
{
    if (nchunks_dev) nchunks = min(nchunks, *nchunks_dev);      // (a plan beyond the launch's room: the host falls back)
    if (blockIdx.x * wpb >= nchunks) return;
    extern __shared__ uint32_t hist[];                       // [nbins + nb] then one u64 for the arg-max
    unsigned long long *best = reinterpret_cast<unsigned long long *>(hist + ((nbins + nb + 1) & ~1u));
    const uint32_t c0 = blockIdx.x * wpb;
    const uint32_t c1 = c0 + wpb < nchunks ? c0 + wpb : nchunks;
    for (uint32_t i = threadIdx.x; i < nbins + nb; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) *best = 0;
    __syncthreads();
    const int64_t lo = nzb[c0], hi = nzb[c1];
    constexpr int kBatch = 8;                                 // loads in flight per thread: the pass is a few round trips, not one per element
    for (int64_t j0 = lo + threadIdx.x; j0 < hi; j0 += (int64_t)blockDim.x * kBatch) {
        uint32_t c[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) { const int64_t j = j0 + (int64_t)u * blockDim.x; c[u] = j < hi ? (uint32_t)cidx[j] : 0xffffffffu; }
#pragma unroll
        for (int u = 0; u < kBatch; u++) if (c[u] != 0xffffffffu) atomicAdd(&hist[c[u] >> binshift], 1u);
    }
    __syncthreads();
    unsigned long long mine = 0;
    for (uint32_t s = threadIdx.x; s < nbins; s += blockDim.x) {
        uint32_t score = 0;
        for (uint32_t t = 0; t < nb; t++) score += hist[s + t];
        const unsigned long long key = ((unsigned long long)score << 32) | (0xffffffffu - s);
        mine = key > mine ? key : mine;
    }
    atomicMax(best, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t s = 0xffffffffu - (uint32_t)(*best & 0xffffffffu);
        uint32_t       wb = ((*best >> 32) ? s << binshift : 0) & ~3u;       // 16-byte aligned for the staging loads
        if (wb + wn > ncols1) wb = ncols1 > wn ? (ncols1 - wn) & ~3u : 0;   // keep the window inside x_ext, 16-byte aligned
        win_base[blockIdx.x] = wb;
    }
}

// Probe for cvr_create's automatic choice of the workgroup layout: over all rows, (a) are the columns of every row in
// ascending order (column phases need it), (b) which share of the non-zeros would an LDS window of x per workgroup serve:
// every 256 consecutive rows histogram their columns over bins of `bin` columns (LDS) and count the fullest two adjacent
// bins -- wherever those lie, so a row shard of a larger matrix (its diagonal shifted) and rectangular matrices are judged
// alike.  Per workgroup b: out[2b] = unsorted flag, out[2b + 1] = non-zeros in those bins (the host adds them up).
__global__ __launch_bounds__(256) void probe_kernel(const int64_t *__restrict__ rp, const int32_t *__restrict__ ci, uint32_t nrows,
                                                    uint32_t bin_shift, uint32_t nbins, unsigned long long *__restrict__ out)
{
    constexpr uint32_t kTile = 8192;                      // elements per pass: their row-start marks take 1 KiB
    extern __shared__ uint32_t hist[];                    // nbins + 1
    __shared__ int64_t  srp[257];
    __shared__ uint32_t starts[kTile / 32];
    __shared__ uint32_t wbest[4], wbad[4];
    unsigned long long near = 0;
    uint32_t           bad = 0;
    const uint32_t     lane = threadIdx.x & 63u;
    for (uint32_t r0 = blockIdx.x * 256; r0 < nrows; r0 += gridDim.x * 256) {
        const uint32_t nr = nrows - r0 < 256 ? nrows - r0 : 256;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i <= nr; i += 256) srp[i] = rp[r0 + i];
        for (uint32_t i = threadIdx.x; i <= nbins; i += 256) hist[i] = 0;
        __syncthreads();
        const int64_t a = srp[0], z = srp[nr];
        for (int64_t t0 = a; t0 < z; t0 += kTile) {
            // first elements of the rows inside this pass, as a bit mask (cheaper than a search per element)
            if (threadIdx.x < kTile / 32) starts[threadIdx.x] = 0;
            __syncthreads();
            if (threadIdx.x < nr) {
                const int64_t p = srp[threadIdx.x] - t0;
                if (p >= 0 && p < (int64_t)kTile && srp[threadIdx.x + 1] > srp[threadIdx.x]) atomicOr(&starts[p >> 5], 1u << (p & 31));
            }
            __syncthreads();
            const int64_t te = z - t0 < (int64_t)kTile ? z : t0 + kTile;
            constexpr int kBatch = 4;                         // loads of four trips in flight (a group of 256 rows is a handful of trips: each used to be a round trip to memory)
            for (int64_t jb0 = t0; jb0 < te; jb0 += 256 * kBatch) {       // (the same trips for every lane: ballots inside)
                int32_t cc[kBatch], cp[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    const int64_t j = jb0 + u * 256 + threadIdx.x;
                    cc[u] = j < te ? ci[j] : 0;
                    cp[u] = j < te && j > a ? ci[j - 1] : 0;
                }
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    const int64_t jb = jb0 + u * 256;
                    if (jb >= te) break;
                    const int64_t  j = jb + threadIdx.x;
                    const bool     valid = j < te;
                    const int32_t  c = cc[u];
                    const uint32_t q = (uint32_t)(j - t0);
                    if (valid && j > a && !((starts[q >> 5] >> (q & 31)) & 1u) && cp[u] > c) bad = 1;
                    // the histogram: lanes of a wavefront that hit the bin of the first pending lane add up in one LDS atomic (the
                    // non-zeros near the diagonal fall into one or two bins); after two such rounds the rest goes one by one
                    const uint32_t     bin = (uint32_t)c >> bin_shift;
                    unsigned long long todo = __ballot(valid);
#pragma unroll 1
                    for (int round = 0; round < 2 && todo; round++) {
                        const int                lead = __ffsll((long long)todo) - 1;
                        const uint32_t           lb = __shfl(bin, lead);
                        const unsigned long long m = __ballot(valid && bin == lb) & todo;
                        if ((int)lane == lead) atomicAdd(&hist[lb], (uint32_t)__popcll(m));
                        todo &= ~m;
                    }
                    if ((todo >> lane) & 1ull) atomicAdd(&hist[bin], 1u);
                }
            }
            __syncthreads();
        }
        uint32_t best = 0;
        for (uint32_t i = threadIdx.x; i < nbins; i += 256) { const uint32_t v = hist[i] + hist[i + 1]; best = v > best ? v : best; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(best, o); best = v > best ? v : best; }
        if (lane == 0) wbest[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) near += std::max(std::max(wbest[0], wbest[1]), std::max(wbest[2], wbest[3]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
    __syncthreads();
    if (lane == 0) wbad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {          // a few workgroups share an output slot (the caller zeroes them)
        const uint32_t slot = blockIdx.x % kProbeBlocks;
        if (wbad[0] | wbad[1] | wbad[2] | wbad[3]) atomicOr(&out[2 * slot], 1ull);
        if (near) atomicAdd(&out[2 * slot + 1], near);
    }
}

// narrow chunks: the smallest column of every chunk, and whether any chunk spans too many columns for 16-bit offsets
__global__ __launch_bounds__(kLanes) void chunk_span_kernel(const int32_t *__restrict__ ci, const int64_t *__restrict__ nzb, uint32_t nchunks,
                                                            uint32_t *__restrict__ cbase, uint32_t *__restrict__ wide)
{
    const uint32_t k = blockIdx.x;
    if (k >= nchunks) return;
    int32_t lo = 0x7fffffff, hi = -1;
    for (int64_t j = nzb[k] + threadIdx.x; j < nzb[k + 1]; j += kLanes) { const int32_t c = ci[j]; lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int32_t a = __shfl_xor(lo, o), b = __shfl_xor(hi, o); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    if (threadIdx.x == 0) {
        cbase[k] = hi >= 0 ? (uint32_t)lo : 0u;
        if (hi >= 0 && (uint32_t)(hi - lo) >= kC16Pad) *wide = 1u;       // (plain store of the same value from every such chunk)
    }
}

// smallest and largest column index of col_idx[n0 .. n1): the range check of cvr_create for CSR arrays that are already
// on the device (the host loop of check_csr otherwise).  minmax[0] = min, minmax[1] = max; the caller initialises both.
__global__ __launch_bounds__(256) void col_range_kernel(const int32_t *__restrict__ ci, long long n0, long long n1, int32_t *minmax)
{
    int32_t lo = 0x7fffffff, hi = (int32_t)0x80000000;
    for (long long j = n0 + (long long)blockIdx.x * 256 + threadIdx.x; j < n1; j += (long long)gridDim.x * 256) {
        const int32_t c = ci[j];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63u) == 0) { atomicMin(&minmax[0], lo); atomicMax(&minmax[1], hi); }
}

// Value-dictionary detection: every workgroup collects the distinct bit patterns of its slice of the values in an LDS
// hash table and merges them into a global one (1024 slots, all-ones = empty; the all-ones pattern itself is reported
// through flags bit 1).  More than kDictMax distinct patterns anywhere -> flags bit 0 (no dictionary).
template <typename B>
__global__ __launch_bounds__(256) void dict_scan_kernel(const B *__restrict__ vals, long long n0, long long n1,
                                                        unsigned long long *__restrict__ table, uint32_t *__restrict__ flags)
{
    constexpr unsigned long long kEmpty = ~0ull;
    __shared__ unsigned long long tab[512];
    __shared__ uint32_t           cnt, over;
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = kEmpty;
    if (threadIdx.x == 0) { cnt = 0; over = flags[0] & 1u; }      // (set: another workgroup already found too many values)
    __syncthreads();
    unsigned long long last = kEmpty;
    constexpr int      kBatch = 8;          // loads in flight per thread (the pass used to be one round trip to memory per element and thread)
    const long long    stride = (long long)gridDim.x * blockDim.x;
    for (long long j0 = n0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; j0 < n1; j0 += stride * kBatch) {
        if (__hip_atomic_load(&over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        B raw[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) { const long long j = j0 + u * stride; raw[u] = j < n1 ? vals[j] : vals[j0]; }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const unsigned long long b = sizeof(B) == 4 && raw[u] == (B)~(B)0 ? kEmpty : (unsigned long long)raw[u];
            if (b == last) continue;
            last = b;
            if (b == kEmpty) { atomicOr(&flags[0], 2u); continue; }
            if (__hip_atomic_load(&over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;      // (too many values already: the table may be full)
            uint32_t hsh = (uint32_t)((b * 0x9E3779B97F4A7C15ull) >> 55);
            for (uint32_t probes = 0; probes < 512; probes++) {
                // a plain read first: once the table holds the matrix's few values nearly every look-up ends here, and lanes reading
                // the same slot are served by one broadcast where the same compare-and-swaps would be executed one after the other
                unsigned long long old = __hip_atomic_load(&tab[hsh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == b) break;
                if (old == kEmpty) old = atomicCAS(&tab[hsh], kEmpty, b);
                if (old == b) break;
                if (old == kEmpty) { if (atomicAdd(&cnt, 1u) + 1 > (uint32_t)kDictMax) __hip_atomic_store(&over, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                hsh = (hsh + 1) & 511u;
                if (__hip_atomic_load(&over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            }
        }
    }
    __syncthreads();
    if (over) { if (threadIdx.x == 0) atomicOr(&flags[0], 1u); return; }
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) {
        const unsigned long long b = tab[i];
        if (b == kEmpty) continue;
        uint32_t hsh = (uint32_t)((b * 0x9E3779B97F4A7C15ull) >> 54);
        for (uint32_t probes = 0; probes < 1024; probes++) {
            // (a load first here too: a thousand workgroups merging the same dozen values would otherwise queue up as
            // compare-and-swaps on a dozen addresses)
            unsigned long long old = __hip_atomic_load(&table[hsh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == b) break;
            if (old == kEmpty) old = atomicCAS(&table[hsh], kEmpty, b);
            if (old == b) break;
            if (old == kEmpty) { if (atomicAdd(&flags[1], 1u) + 1 > (uint32_t)kDictMax) atomicOr(&flags[0], 1u); break; }
            hsh = (hsh + 1) & 1023u;
            if (flags[0] & 1u) break;
        }
    }
}

}  // namespace

hipError_t launch_probe(const int64_t *rp, const int32_t *ci, int64_t nrows, int64_t ncols, uint32_t half, unsigned long long *out2, hipStream_t st, bool out_zeroed)
{
    if (nrows <= 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<int64_t>(16384, (nrows + 255) / 256);      // one group of 256 rows each (more rows: several)
    uint32_t       bin_shift = 0;                          // bins of `half` columns (a power of two), at most 8192 of them
    while ((2u << bin_shift) <= half) bin_shift++;
    while (((uint64_t)ncols >> bin_shift) + 1 > 8192) bin_shift++;
    const uint32_t nbins = (uint32_t)(((uint64_t)ncols >> bin_shift) + 1);
    if (!out_zeroed) {
        const hipError_t e = hipMemsetAsync(out2, 0, sizeof(unsigned long long) * 2 * kProbeBlocks, st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), sizeof(uint32_t) * (nbins + 1), st, rp, ci, (uint32_t)nrows, bin_shift, nbins, out2);
    return hipGetLastError();
}

hipError_t launch_chunk_span(const DeviceImage &img, const DeviceCsr &csr, uint32_t *cbase, uint32_t *wide, hipStream_t st)
{
    if (img.nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(chunk_span_kernel, dim3(img.nchunks), dim3(kLanes), 0, st, csr.col_idx, csr.nz_begin, img.nchunks, cbase, wide);
    return hipGetLastError();
}

// out[0] = rp[0], out[1] = rp[nrows], out[2] = the first row whose successor's pointer is smaller (-1: none); out is written whole
__global__ __launch_bounds__(256) void rows_check_kernel(const long long *__restrict__ rp, long long nrows, long long *__restrict__ out)
{
    long long bad = 0x7fffffffffffffffll;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < nrows; r += (long long)gridDim.x * 256)
        if (rp[r + 1] < rp[r]) { bad = r; break; }
    if (bad != 0x7fffffffffffffffll) atomicMin(reinterpret_cast<unsigned long long *>(out + 2), (unsigned long long)bad);
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = rp[0]; out[1] = rp[nrows]; }
}

hipError_t launch_rows_check(const int64_t *rp, int64_t nrows, long long *out3, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(out3, 0xff, sizeof(long long) * 3, st);      // (all ones: "no bad row" for the unsigned minimum, read back as -1)
    if (e != hipSuccess) return e;
    const uint32_t blocks = (uint32_t)std::min<int64_t>(4096, (nrows + 255) / 256);
    hipLaunchKernelGGL(rows_check_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const long long *>(rp), (long long)nrows, out3);
    return hipGetLastError();
}

hipError_t launch_col_range(const int32_t *ci, int64_t n0, int64_t n1, int32_t *minmax, hipStream_t st)
{
    if (n1 <= n0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<int64_t>(2048, (n1 - n0 + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(col_range_kernel, dim3(blocks), dim3(256), 0, st, ci, (long long)n0, (long long)n1, minmax);
    return hipGetLastError();
}

hipError_t launch_dict_scan(const void *vals, int64_t n0, int64_t n1, bool f32, unsigned long long *table, uint32_t *flags, hipStream_t st)
{
    if (n1 <= n0) return hipSuccess;
    const int64_t n = n1 - n0;
    const uint32_t blocks = (uint32_t)std::min<int64_t>(2048, (n + 256 * 16 - 1) / (256 * 16));
    if (f32) hipLaunchKernelGGL(dict_scan_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, static_cast<const uint32_t *>(vals), (long long)n0, (long long)n1, table, flags);
    else hipLaunchKernelGGL(dict_scan_kernel<uint64_t>, dim3(blocks), dim3(256), 0, st, static_cast<const uint64_t *>(vals), (long long)n0, (long long)n1, table, flags);
    return hipGetLastError();
}

// elements of a chunk whose column indices and row-start flags (5 bytes each) fit the LDS beside the row pieces (at most ystage - 1 rows, 8 bytes each)
static uint32_t seg_lds_cols(const DeviceImage &img)
{
    const size_t cap = (size_t)kLanes * img.S, room = (kLdsBytes - 8 * (size_t)img.ystage - 1024) / 5;      // (1 KiB: the kernel's static tables)
    return (uint32_t)std::min(cap, room);
}
static size_t seg_lds_bytes(const DeviceImage &img) { return 8 * (size_t)img.ystage + 5 * (size_t)seg_lds_cols(img) + 16; }

// LDS of seg_scan_kernel for a chunk of `cap` elements and at most ystage - 1 rows (0: such chunks do not fit, seg_build_kernel takes them)
static size_t seg_scan_lds_bytes(const DeviceImage &img)
{
    const size_t cap = (size_t)kLanes * img.S;
    if (cap > (size_t)kSegRunMax * kSegThreads || img.phases > 64) return 0;
    const size_t bytes = 4 * (cap / 64 + 2) + 4 * cap + 2 * (size_t)img.phases * kSegThreads + 2 * cap + 2 * cap + cap + cap + 64;
    return bytes <= kLdsBytes - 1024 ? bytes : 0;
}

bool seg_table_packed_ok(const DeviceImage &img) { return seg_scan_lds_bytes(img) != 0 && !cvr::debug_env("seg_by_rows"); }

hipError_t launch_seg_total(const SegTable &st, uint32_t nchunks, const uint32_t *nchunks_dev, uint32_t *total_out, hipStream_t s)
{
    hipLaunchKernelGGL(seg_total_kernel, dim3(1), dim3(1024), 0, s, st.cnt, nchunks, nchunks_dev, total_out);
    return hipGetLastError();
}

hipError_t launch_seg_build(const DeviceImage &img, const DeviceCsr &csr, SegTable &st, hipStream_t s, const uint32_t *nchunks_dev, bool with_total)
{
    if (img.nchunks == 0) return hipSuccess;
    const bool by_rows = cvr::debug_env("seg_by_rows") != nullptr;      // (diagnostics: the search-based kernel)
    if (const size_t lds = by_rows ? 0 : seg_scan_lds_bytes(img)) {
        unsigned long long *dbg = nullptr;
        if (cvr::debug_env("seg_clocks") && hipMalloc(&dbg, sizeof(unsigned long long) * 16 * 64) != hipSuccess) dbg = nullptr;
        if (dbg) fprintf(stderr, "[seg_scan] %u chunks, S %d, %u phases, ystage %u, LDS %zu bytes per workgroup\n", img.nchunks, img.S, img.phases, img.ystage, lds);
        hipLaunchKernelGGL(seg_scan_kernel, dim3(img.nchunks), dim3(kSegThreads), lds, s, csr.row_ptr, csr.col_idx, csr.nz_begin, csr.pad_cnt, img.desc, img.desc2,
                           img.nchunks, nchunks_dev, img.phase_width, img.phases, (uint32_t)(kLanes * img.S), st.cnt, reinterpret_cast<uint2 *>(st.begin), st.flags,
                           img.piece_max ? (uint32_t)__builtin_ctz(img.piece_max) : 31u, dbg);
        st.packed = true;
        if (dbg) {          // CVR_SEG_CLOCKS: 100-MHz time stamps of every 256th workgroup's stages, on stderr
            unsigned long long h[16 * 64] = {};
            if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                unsigned long long t0 = ~0ull;
                for (uint32_t g = 0; g * 256 < img.nchunks && g < 64; g++) t0 = std::min(t0, h[g * 16]);
                for (uint32_t g = 0; g * 256 < img.nchunks && g < 64; g++) {
                    fprintf(stderr, "[seg_scan] workgroup %5u: start %7.2f us; stages (us)", g * 256, (double)(h[g * 16] - t0) * 0.01);
                    for (int i = 1; i <= 6; i++) fprintf(stderr, " %6.2f", (double)(h[g * 16 + i] - h[g * 16 + i - 1]) * 0.01);
                    fprintf(stderr, "\n");
                }
            }
            (void)hipFree(dbg);
        }
        if (with_total) hipLaunchKernelGGL(seg_total_kernel, dim3(1), dim3(1024), 0, s, st.cnt, img.nchunks, nchunks_dev, st.cnt + img.nchunks);
        return hipGetLastError();
    }
    st.packed = false;
    hipLaunchKernelGGL(seg_build_kernel, dim3(img.nchunks), dim3(kLanes * kSegWaves), seg_lds_bytes(img), s, csr.row_ptr, csr.col_idx, csr.nz_begin, csr.pad_cnt, img.desc,
                       img.desc2, img.nchunks, img.phase_width, img.phases, (uint32_t)(kLanes * img.S), st.cnt, st.begin, st.len, st.row, st.flags, seg_lds_cols(img), img.piece_max ? (uint32_t)__builtin_ctz(img.piece_max) : 31u);      // (piece_max is a power of two: cvr_layout)
    if (with_total) hipLaunchKernelGGL(seg_total_kernel, dim3(1), dim3(1024), 0, s, st.cnt, img.nchunks, nchunks_dev, st.cnt + img.nchunks);
    return hipGetLastError();
}

hipError_t launch_dict_codes(const void *vals, int64_t n0, int64_t n1, bool f32, const void *dict, uint32_t ndict, uint8_t *codes, uint32_t *err_flag, hipStream_t st)
{
    if (n1 <= n0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<int64_t>(4096, (n1 - n0 + 256 * 16 - 1) / (256 * 16));
    if (f32) hipLaunchKernelGGL(dict_codes_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float *>(vals), (long long)n0, (long long)n1, static_cast<const float *>(dict), ndict, codes, err_flag);
    else hipLaunchKernelGGL(dict_codes_kernel<double>, dim3(blocks), dim3(256), 0, st, static_cast<const double *>(vals), (long long)n0, (long long)n1, static_cast<const double *>(dict), ndict, codes, err_flag);
    return hipGetLastError();
}

hipError_t launch_convert(const DeviceImage &img, const DeviceCsr &csr, uint32_t *err_flag, hipStream_t st, const SegTable *seg, const uint32_t *nchunks_dev)
{
    if (img.nchunks == 0) return hipSuccess;
    unsigned long long *dbg = nullptr;
    if (cvr::debug_env("convert_clocks") && hipMalloc(&dbg, sizeof(unsigned long long) * 16 * 64) == hipSuccess) {
        (void)hipMemset(dbg, 0, sizeof(unsigned long long) * 16 * 64);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_dbg), &dbg, sizeof(dbg));
    }
    struct DbgPrint {
        unsigned long long *dbg; hipStream_t st; uint32_t n;
        ~DbgPrint()
        {
            if (!dbg) return;
            unsigned long long h[16 * 64] = {}, *none = nullptr;
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                unsigned long long t0 = ~0ull;
                for (uint32_t g = 0; g * 256 < n && g < 64; g++) if (h[g * 16]) t0 = std::min(t0, h[g * 16]);
                for (uint32_t g = 0; g * 256 < n && g < 64; g++) {
                    if (!h[g * 16]) continue;
                    fprintf(stderr, "[convert] chunk %5u: start %7.2f us; staging, groups 0-3 up to the hand-out of the fourth, its loads / gathers + stores, the other groups (us)", g * 256, (double)(h[g * 16] - t0) * 0.01);
                    for (int i = 1; i <= 4; i++) fprintf(stderr, " %6.2f", (double)(h[g * 16 + i] - h[g * 16 + i - 1]) * 0.01);
                    fprintf(stderr, "\n");
                }
            }
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_dbg), &none, sizeof(none));
            (void)hipFree(dbg);
        }
    } dbg_print{dbg, st, img.nchunks};
    // the LDS-staged kernel: when the values come as dictionary codes (csr.codes), or with CVR_CONVERT_LDS=1 (with the values themselves it
    // is not faster on the web-Google shape: DESIGN.md section 5.11)
    if (seg && seg->packed && !img.c16 && !img.hub_n && ((csr.codes && img.dict) || cvr::debug_env("convert_lds"))) {
        const size_t capl = (size_t)kLanes * img.S, vb = img.dict ? 1 : (img.f32 ? 4 : 8), db = img.dict ? (size_t)((img.ndict + 3u) & ~3u) * (img.f32 ? 4 : 8) : 0;
        const size_t lds = 16 * (size_t)kRingHalf + db + capl * (4 + vb) + 16;
        if (lds <= (48u << 10)) {           // at least three chunks per CU
            const uint2 *pk = reinterpret_cast<const uint2 *>(seg->begin);
#define CVR_CONVERT_LDS(T, DI, TG)                                                                                                                         \
    hipLaunchKernelGGL((convert_lds_kernel<T, DI, TG>), dim3(img.nchunks), dim3(kLanes), lds, st, csr.col_idx, static_cast<const T *>(csr.vals), csr.nz_begin,  \
                       img.desc, img.desc2, img.stream, img.target, err_flag, img.G, img.nchunks, nchunks_dev, img.pad_col,                                      \
                       static_cast<const T *>(img.dict), img.ndict, pk, img.col_bits, seg->flags, img.dict ? csr.codes : nullptr)
#define CVR_CONVERT_LDS_TG(T, DI) do { if (img.tag16) CVR_CONVERT_LDS(T, DI, true); else CVR_CONVERT_LDS(T, DI, false); } while (0)
            if (img.f32) { if (img.dict) CVR_CONVERT_LDS_TG(float, true); else CVR_CONVERT_LDS_TG(float, false); }
            else         { if (img.dict) CVR_CONVERT_LDS_TG(double, true); else CVR_CONVERT_LDS_TG(double, false); }
#undef CVR_CONVERT_LDS_TG
#undef CVR_CONVERT_LDS
            return hipGetLastError();
        }
    }
    const uint32_t blocks = (img.nchunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3     grid(blocks), block(kLanes * kWavesPerBlock);
    // the feed table of a chunk in LDS (convert_kernel, STAGE) when eight chunks' tables fit a CU (16-bit positions: 64 S < 65 535)
    const size_t cap = (size_t)kLanes * img.S;
    size_t       per = seg ? 6 * cap : 4 * (cap + 2);
    per = (per + 15) & ~(size_t)15;
    const bool direct = cvr::debug_env("convert_direct") != nullptr;       // (diagnostics: the unstaged kernel)
    const bool   stage = !img.c16 && cap < 65535 && per * kWavesPerBlock <= (20u << 10) && !direct;
    const size_t lds = stage ? per * kWavesPerBlock : 0;
#define CVR_CONVERT_ARGS(T)                                                                                            \
    csr.row_ptr, csr.col_idx, static_cast<const T *>(csr.vals), csr.nz_begin, csr.pad_cnt, img.desc, img.stream, img.target, err_flag, img.G, img.nchunks, img.pad_col, \
    static_cast<const T *>(img.dict), img.ndict, img.desc2, seg ? seg->begin : nullptr, seg ? seg->len : nullptr, seg ? seg->row : nullptr, img.col_bits,           \
    seg ? seg->flags : nullptr, img.hub_n ? img.hub_index : nullptr, img.hub_bitmap, img.cbase, img.hub_n, (uint32_t)per, seg && seg->packed ? 1u : 0u, nchunks_dev
#define CVR_CONVERT(T, DI, SG, SM)                                                                                     \
    do {                                                                                                               \
        if (SG && img.tag16) hipLaunchKernelGGL((convert_kernel<T, DI, SG, false, SM, SG>), grid, block, lds, st, CVR_CONVERT_ARGS(T)); \
        else hipLaunchKernelGGL((convert_kernel<T, DI, SG, false, SM, false>), grid, block, lds, st, CVR_CONVERT_ARGS(T)); \
    } while (0)
#define CVR_CONVERT_SM(T, DI, SG) do { if (stage) CVR_CONVERT(T, DI, SG, true); else CVR_CONVERT(T, DI, SG, false); } while (0)
#define CVR_CONVERT_SG(T, DI) do { if (seg) CVR_CONVERT_SM(T, DI, true); else CVR_CONVERT_SM(T, DI, false); } while (0)
    if (img.c16 && !img.dict && !seg) {      // narrow chunks: 16-bit column offsets
        if (img.f32) hipLaunchKernelGGL((convert_kernel<float, false, false, true, false>), grid, block, 0, st, csr.row_ptr, csr.col_idx, static_cast<const float *>(csr.vals),
                                        csr.nz_begin, csr.pad_cnt, img.desc, img.stream, img.target, err_flag, img.G, img.nchunks, img.pad_col,
                                        (const float *)nullptr, 0u, img.desc2, (const int64_t *)nullptr, (const uint32_t *)nullptr, (const uint16_t *)nullptr, img.col_bits,
                                        (const uint32_t *)nullptr, (const int32_t *)nullptr, (const uint32_t *)nullptr, img.cbase, 0u, 0u, 0u, nchunks_dev);
        else hipLaunchKernelGGL((convert_kernel<double, false, false, true, false>), grid, block, 0, st, csr.row_ptr, csr.col_idx, static_cast<const double *>(csr.vals),
                                csr.nz_begin, csr.pad_cnt, img.desc, img.stream, img.target, err_flag, img.G, img.nchunks, img.pad_col,
                                (const double *)nullptr, 0u, img.desc2, (const int64_t *)nullptr, (const uint32_t *)nullptr, (const uint16_t *)nullptr, img.col_bits,
                                (const uint32_t *)nullptr, (const int32_t *)nullptr, (const uint32_t *)nullptr, img.cbase, 0u, 0u, 0u, nchunks_dev);
        return hipGetLastError();
    }
    if (img.f32) { if (img.dict) CVR_CONVERT_SG(float, true); else CVR_CONVERT_SG(float, false); }
    else         { if (img.dict) CVR_CONVERT_SG(double, true); else CVR_CONVERT_SG(double, false); }
#undef CVR_CONVERT_SG
#undef CVR_CONVERT_SM
#undef CVR_CONVERT_ARGS
#undef CVR_CONVERT
    return hipGetLastError();
}

}  // namespace cvr

namespace cvr {

hipError_t launch_window(const DeviceImage &img, const DeviceCsr &csr, hipStream_t st, const uint32_t *nchunks_dev)
{
    if (img.nchunks == 0 || img.win_elems == 0) return hipSuccess;
    const uint32_t wpb = img.wpb > 1 ? img.wpb : 1;
    const uint32_t blocks = (img.nchunks + wpb - 1) / wpb;
    const uint32_t ncols1 = img.pad_col + 1;
    uint32_t       binshift = 0;
    while ((1u << binshift) * 16 < img.win_elems) binshift++;             // bins of >= window/16 columns ...
    while (((uint64_t)ncols1 >> binshift) + 1 > 8192) binshift++;         // ... and at most 8192 of them
    const uint32_t nbins = (uint32_t)(((uint64_t)ncols1 + (1u << binshift) - 1) >> binshift);
    uint32_t       nb = img.win_elems >> binshift;
    if (nb == 0) nb = 1;
    const size_t lds = sizeof(uint32_t) * ((size_t)((nbins + nb + 1) & ~1u)) + 16;
    hipLaunchKernelGGL(window_kernel, dim3(blocks), dim3(1024), lds, st, csr.col_idx, csr.nz_begin, img.nchunks, ncols1,
                       img.win_elems, binshift, nbins, nb, img.win_base, wpb, nchunks_dev);
    return hipGetLastError();
}

// (cvr_create's warm-up thread: asking for a kernel's attributes makes the runtime load this file's code object, which the first launch would
// otherwise wait for)
void touch_convert_kernels()
{
    hipFuncAttributes a;
    for (const void *k : {reinterpret_cast<const void *>(&seg_scan_kernel), reinterpret_cast<const void *>(&seg_total_kernel), reinterpret_cast<const void *>(&window_kernel),
                          reinterpret_cast<const void *>(&probe_kernel)})
        (void)hipFuncGetAttributes(&a, k);
    (void)hipGetLastError();
}

}  // namespace cvr

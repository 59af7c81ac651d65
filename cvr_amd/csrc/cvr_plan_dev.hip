// cvr_plan_dev.hip -- the chunk planner on the device: the same plan as cvr_plan.cpp (chunk for chunk; tests compare them),
// computed from a row_ptr that is already in device memory, so that neither the row pointers (8 B per row) nor the walk over
// them (one add, compare and branch per row) have to pass through the host.
//
// Counterpart of the reference's per-thread partition and its first/last-row binary searches (spmv.cpp:584-667): there every
// thread searches its own two rows; here every ROW searches where a chunk that starts at it would end, and one wavefront per
// row block (kPlanRowBlock rows, the planner's restart interval) then follows those jumps from the block's first row.
//
//   tile_kernel   empty rows per tile of 1 024 rows
//   q_kernel      Q[r] = slots in front of row r inside its row block (a row owns max(1, nnz) slots), 32 bits
//   jump_kernel   J[r] = e - r for the last row boundary e with Q[e] - Q[r] <= 64 S (and <= max_rows rows), bit 15: the row at
//                 e is longer than the split threshold, i.e. the chunk is filled with its first piece (binary search, the
//                 lanes of a wavefront probe neighbouring words)
//   walk_kernel   per row block: J in LDS (128 KiB), wavefront 0 follows it; rows cut over chunks (rare) go through Q with
//                 a 64-way search.  Emits one start record (row, offset inside the row, cut rows so far) per chunk
//   emit_kernel   chunk records and the list of cut rows, numbered across the blocks
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "cvr_kernels.h"
#include "cvr_plan.h"

namespace cvr {
namespace {

constexpr int      kTileRows = 1024;
constexpr int      kTilesPerBlock = (int)(kPlanRowBlock / kTileRows);
constexpr uint32_t kJumpCut = 0x8000u;
static_assert(kPlanRowBlock % kTileRows == 0 && kPlanRowBlock * sizeof(uint16_t) <= 128 * 1024, "a row block's jump table fits the LDS");

struct Start { uint32_t row, off, open; };      // chunk start: global row, slots of that row already placed, cut rows opened before it in the block

struct ChunkRec {                                // = cvr::Chunk, byte for byte
    long long nz_begin, row_first, nrows_in, nseg, pad_cnt;
    uint8_t   head_shared, tail_shared, fill[6];
};
static_assert(sizeof(ChunkRec) == sizeof(Chunk) && sizeof(Chunk) == 48, "device chunk record = host Chunk");

struct PlanArgs {
    const long long *rp;
    long long        nrows;
    uint32_t         nblocks, cap, thr, max_rows;      // max_rows: 0xffffffff = no cap
    uint32_t        *tile_empty;                       // [ntiles]
    uint32_t        *Q;                                // [nrows + nblocks]: block b's entries 0 .. n_b at b * (kPlanRowBlock + 1)
    uint16_t        *J;                                // [nrows]
    Start           *starts;                           // [bound]
    uint32_t        *counts;                           // [2 * nblocks]: chunks, cut rows per block
    ChunkRec        *chunks;                           // [bound]
    Shared          *shared;                           // [bound]
    unsigned long long *totals;                        // chunks, cut rows, flags (1: a block holds 2^31 slots or more), most rows in a chunk
    // optional (cvr_fused.hip): the per-chunk tables of the image written straight from the plan (what plan_part derives on the host)
    uint4           *odesc;                            // [room] {row_first, nseg, head_dest, last_dest}
    uint2           *odesc2;                           // [room] {0, rows}
    uint32_t        *opad;                             // [room]
    long long       *onzb;                             // [room + 1]
    uint32_t         oroom;                            // chunks the tables have room for (more: totals[2] |= 2)
    uint32_t         ophased;                          // last_dest = that of the last ROW (column phases) instead of the last segment
    long long        nz_end;
};

__device__ inline uint32_t rows_of(const PlanArgs &a, uint32_t b) { return (uint32_t)min((long long)kPlanRowBlock, a.nrows - (long long)b * kPlanRowBlock); }
__device__ inline size_t   qbase(uint32_t b) { return (size_t)b * (kPlanRowBlock + 1); }
// room for block b's chunks: every chunk but the block's last is full, or more than half full (the row that did not fit is
// at most thr <= cap/2 long), or holds max_rows rows
__device__ inline uint32_t bound_of(const PlanArgs &a, uint32_t slots, uint32_t rows) { return 2u * (slots / a.cap) + (a.max_rows != 0xffffffffu ? rows / a.max_rows : 0u) + 3u; }

__global__ __launch_bounds__(kTileRows) void tile_kernel(PlanArgs a)
{
    __shared__ uint32_t wsum[kTileRows / 64];
    const long long r = (long long)blockIdx.x * kTileRows + threadIdx.x;
    const bool      empty = r < a.nrows && a.rp[r + 1] == a.rp[r];
    const uint32_t  n = (uint32_t)__popcll(__ballot(empty));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < kTileRows / 64; w++) s += wsum[w];
        a.tile_empty[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kTileRows) void q_kernel(PlanArgs a)
{
    __shared__ uint32_t wsum[kTileRows / 64];
    __shared__ uint32_t before;
    const uint32_t  b = blockIdx.x / kTilesPerBlock, t = blockIdx.x % kTilesPerBlock;
    const long long r0 = (long long)b * kPlanRowBlock, r = (long long)blockIdx.x * kTileRows + threadIdx.x;
    const long long base = a.rp[r0];
    // empty rows of the block's earlier tiles
    if (threadIdx.x < 64) {
        uint32_t v = threadIdx.x < t ? a.tile_empty[b * kTilesPerBlock + threadIdx.x] : 0u;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (threadIdx.x == 0) before = v;
    }
    const long long          here = r <= a.nrows ? a.rp[r] : 0;
    const bool               empty = r < a.nrows && a.rp[r + 1] == here;
    const unsigned long long m = __ballot(empty);
    const uint32_t           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t e = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; w++) e += wsum[w];
    if (r < a.nrows) {
        const long long q = here - base + (long long)e;
        a.Q[qbase(b) + (size_t)(r - r0)] = (uint32_t)q;
        // the entry behind the block's last row
        if (r + 1 == a.nrows || r + 1 == r0 + kPlanRowBlock) {
            const long long qe = a.rp[r + 1] - base + (long long)e + (empty ? 1 : 0);
            a.Q[qbase(b) + (size_t)(r + 1 - r0)] = (uint32_t)qe;
            if (qe >= (1ll << 31)) atomicOr(a.totals + 2, 1ull);
        }
    }
}

__global__ __launch_bounds__(256) void jump_kernel(PlanArgs a)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.nrows) return;
    const uint32_t  b = (uint32_t)(r / kPlanRowBlock), i = (uint32_t)(r % kPlanRowBlock), nb = rows_of(a, b);
    const uint32_t *Q = a.Q + qbase(b);
    const uint32_t  lim = (uint32_t)min((unsigned long long)nb, (unsigned long long)i + a.max_rows);
    const uint32_t  q0 = Q[i], target = q0 + a.cap;
    uint32_t        lo = i, hi = min(lim, i + a.cap);       // the last e in [i, lim] with Q[e] <= target (every row owns a slot: e - i <= cap)
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (Q[mid] <= target) lo = mid; else hi = mid - 1;
    }
    const bool cut = lo < lim && Q[lo] - q0 < a.cap && Q[lo + 1] - Q[lo] > a.thr;
    a.J[r] = (uint16_t)((lo - i) | (cut ? kJumpCut : 0u));
}

// the last e in [lo, hi] with Q[e] <= target (Q ascending, Q[lo] <= target): the wavefront probes 64 positions per round
__device__ inline uint32_t wave_upper(const uint32_t *Q, uint32_t lo, uint32_t hi, uint32_t target, uint32_t lane)
{
    while (lo < hi) {
        const uint32_t step = (hi - lo + 63) / 64;
        const uint32_t p = min(lo + (lane + 1) * step, hi);
        const uint32_t cnt = (uint32_t)__popcll(__ballot(Q[p] <= target));
        if (cnt == 64) return hi;
        const uint32_t nlo = lo + cnt * step;
        hi = min(hi, lo + (cnt + 1) * step) - 1;
        lo = nlo;
    }
    return lo;
}

__global__ __launch_bounds__(256) void walk_kernel(PlanArgs a)
{
    extern __shared__ uint16_t jl[];
    __shared__ uint32_t red[4];
    const uint32_t  b = blockIdx.x, nb = rows_of(a, b), lane = threadIdx.x & 63;
    const uint32_t *Q = a.Q + qbase(b);
    const long long r0 = (long long)b * kPlanRowBlock;
    {   // the block's jump table, 16 bytes per load (the table starts at a multiple of 128 KiB)
        const uint4 *src = reinterpret_cast<const uint4 *>(a.J + r0);
        uint4       *dst = reinterpret_cast<uint4 *>(jl);
        const uint32_t n16 = nb / 8;
        for (uint32_t k = threadIdx.x; k < n16; k += 256) dst[k] = src[k];
        for (uint32_t k = n16 * 8 + threadIdx.x; k < nb; k += 256) jl[k] = a.J[r0 + k];
    }
    // where this block's start records go: behind the room of the blocks in front of it
    uint32_t mine = 0;
    for (uint32_t c = threadIdx.x; c < b; c += 256) mine += bound_of(a, a.Q[qbase(c) + rows_of(a, c)], rows_of(a, c));
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if (lane == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (a.totals[2] & 1ull) { if (lane == 0) { a.counts[2 * b] = 0; a.counts[2 * b + 1] = 0; } return; }      // a block beyond 32-bit slot positions: the host plans
    Start         *out = a.starts + (size_t)(red[0] + red[1] + red[2] + red[3]);
    const uint32_t cap = a.cap;
    uint32_t       r = 0, off = 0, count = 0, open = 0;
    while (r < nb) {
        if (lane == 0) out[count] = Start{(uint32_t)(r0 + r), off, open};
        count++;
        if (off == 0) {
            const uint32_t j = jl[r], e = r + (j & (kJumpCut - 1u));
            if (j & kJumpCut) { off = cap - (Q[e] - Q[r]); open++; }
            r = e;
            continue;
        }
        const uint32_t rem = Q[r + 1] - Q[r] - off;          // what is left of the row this chunk begins with
        if (rem > cap) { off += cap; continue; }              // a chunk full of it
        if (rem == cap) { r++; off = 0; continue; }
        const uint32_t lim = (uint32_t)min((unsigned long long)nb, (unsigned long long)r + a.max_rows);
        if (r + 1 >= lim) { r++; off = 0; continue; }
        const uint32_t room = cap - rem;
        const uint32_t e = wave_upper(Q, r + 1, min(lim, r + 1 + room), Q[r + 1] + room, lane);
        const uint32_t used = rem + (Q[e] - Q[r + 1]);
        if (e < lim && used < cap && Q[e + 1] - Q[e] > a.thr) { off = cap - used; open++; }
        else off = 0;
        r = e;
    }
    if (lane == 0) { a.counts[2 * b] = count; a.counts[2 * b + 1] = open; }
}

__global__ __launch_bounds__(256) void emit_kernel(PlanArgs a)
{
    __shared__ uint32_t red[3][4];
    const uint32_t b = blockIdx.x, nb = rows_of(a, b), lane = threadIdx.x & 63;
    uint32_t       room = 0, kb = 0, sb = 0;
    for (uint32_t c = threadIdx.x; c < b; c += 256) {
        room += bound_of(a, a.Q[qbase(c) + rows_of(a, c)], rows_of(a, c));
        kb += a.counts[2 * c];
        sb += a.counts[2 * c + 1];
    }
    for (int o = 32; o > 0; o >>= 1) { room += __shfl_down(room, o); kb += __shfl_down(kb, o); sb += __shfl_down(sb, o); }
    if (lane == 0) { red[0][threadIdx.x >> 6] = room; red[1][threadIdx.x >> 6] = kb; red[2][threadIdx.x >> 6] = sb; }
    __syncthreads();
    room = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    kb = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    sb = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    const uint32_t  count = a.counts[2 * b];
    const Start    *st = a.starts + room;
    const uint32_t *Q = a.Q + qbase(b);
    const long long r0 = (long long)b * kPlanRowBlock;
    uint32_t        most = 0;
    for (uint32_t i = threadIdx.x; i < count; i += 256) {
        const Start     s = st[i];
        const Start     n = i + 1 < count ? st[i + 1] : Start{(uint32_t)(r0 + nb), 0u, 0u};
        const uint32_t  lr = (uint32_t)(s.row - r0), ln = (uint32_t)(n.row - r0);
        const long long used = ((long long)Q[ln] + n.off) - ((long long)Q[lr] + s.off);
        ChunkRec        c;
        c.nz_begin = a.rp[s.row] + s.off;
        c.row_first = s.row;
        c.head_shared = s.off > 0;
        c.tail_shared = n.off > 0;
        const long long last = c.tail_shared ? (long long)n.row : (long long)n.row - 1;
        c.nrows_in = last - (long long)s.row + 1;
        c.pad_cnt = (long long)a.cap - used;
        c.nseg = c.nrows_in + (c.pad_cnt > 0 ? 1 : 0);
        for (int f = 0; f < 6; f++) c.fill[f] = 0;
        a.chunks[(size_t)kb + i] = c;
        most = max(most, 2u * (uint32_t)c.nrows_in + (c.pad_cnt > 0 ? 1u : 0u));      // (the most rows of a chunk, and whether such a chunk has a pad segment: 2 rows + flag)
        if (a.odesc && kb + i < a.oroom) {
            const uint32_t kk = kb + i, nr = (uint32_t)a.nrows;
            // where segment q writes: a row begun earlier -> carry_head(k); a row continued later -> carry_tail(k); the pad segment -> dump
            auto dest = [&](long long q) -> uint32_t {
                if (q >= c.nrows_in) return nr;
                if (q == 0 && c.head_shared) return nr + 1u + 2u * kk;
                if (q == c.nrows_in - 1 && c.tail_shared) return nr + 1u + 2u * kk + 1u;
                return (uint32_t)(c.row_first + q);
            };
            a.odesc[kk] = uint4{(uint32_t)c.row_first, (uint32_t)c.nseg, dest(0), dest(a.ophased ? c.nrows_in - 1 : c.nseg - 1)};
            if (a.odesc2) a.odesc2[kk] = uint2{0u, (uint32_t)c.nrows_in};
            a.opad[kk] = (uint32_t)c.pad_cnt;
            a.onzb[kk] = c.nz_begin;
        } else if (a.odesc && kb + i == a.oroom) a.onzb[a.oroom] = c.nz_begin;      // (a plan beyond the room: the entry that ends the last chunk there is)

        // a row whose first piece ends this chunk: it ends in the chunk behind those it fills completely
        if (c.tail_shared && !(c.head_shared && last == (long long)s.row)) {
            const uint32_t rem = Q[ln + 1] - Q[ln] - n.off;
            const uint32_t middle = rem > a.cap ? (rem + a.cap - 1) / a.cap - 1 : 0u;
            a.shared[(size_t)sb + s.open] = Shared{(int64_t)n.row, (int64_t)kb + i, (int64_t)kb + i + 1 + middle};
        }
    }
    if (b + 1 == gridDim.x && threadIdx.x == 0) {
        a.totals[0] = (unsigned long long)kb + count; a.totals[1] = (unsigned long long)sb + a.counts[2 * b + 1];
        if (a.odesc) {
            if (kb + count <= a.oroom) a.onzb[kb + count] = a.nz_end;
            else atomicOr(a.totals + 2, 2ull);
        }
    }
    if (a.odesc) {          // the most rows any chunk holds (sizes the SpMV kernel's row accumulators)
        for (int o = 32; o > 0; o >>= 1) most = max(most, (uint32_t)__shfl_down(most, o));
        if (lane == 0 && most) atomicMax(a.totals + 3, (unsigned long long)most);
    }
}

__global__ __launch_bounds__(256) void max_row_kernel(const long long *__restrict__ rp, long long nrows, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long red[4];
    unsigned long long m = 0;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < nrows; r += (long long)gridDim.x * 256) m = max(m, (unsigned long long)(rp[r + 1] - rp[r]));
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned long long)__shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}

__global__ __launch_bounds__(256) void shift_rows_kernel(const long long *__restrict__ src, long long n, long long base, long long *__restrict__ dst)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i] - base;
}

// block_off[b] = the first of the n ascending row numbers that is >= b * kCombineRows (b = 0 .. nblocks)
__global__ __launch_bounds__(256) void block_off_kernel(const uint32_t *__restrict__ rows, uint32_t n, uint32_t nblocks, uint32_t *__restrict__ out)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b > nblocks) return;
    const unsigned long long lim = (unsigned long long)b * kCombineRows;
    uint32_t                 lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rows[mid] < lim) lo = mid + 1; else hi = mid;
    }
    out[b] = lo;
}

}  // namespace

// (the jump table holds, per row, how many ROWS a chunk starting there spans, in 15 bits: at most min(slots of a chunk, the row cap))
bool plan_on_device_ok(int32_t S, int64_t max_rows) { return (int64_t)kLanes * S < (int64_t)kJumpCut || (max_rows > 0 && max_rows < (int64_t)kJumpCut); }

// room of the record arrays (an upper bound of the chunks) and bytes of device scratch a plan of these rows needs
int64_t plan_bound_device(int64_t nrows, int64_t nz_end, int32_t S, int64_t max_rows)
{
    const int64_t cap = (int64_t)kLanes * S, nblocks = (nrows + kPlanRowBlock - 1) / kPlanRowBlock;
    return 2 * ((nz_end + nrows) / cap) + (max_rows > 0 ? nrows / max_rows : 0) + 3 * nblocks;
}
size_t plan_scratch_bytes(int64_t nrows, int64_t nz_end, int32_t S, int64_t max_rows)
{
    const int64_t nblocks = (nrows + kPlanRowBlock - 1) / kPlanRowBlock, ntiles = (nrows + kTileRows - 1) / kTileRows, bound = plan_bound_device(nrows, nz_end, S, max_rows);
    auto          up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(4 * (size_t)ntiles) + up(4 * (size_t)(nrows + nblocks)) + up(2 * (size_t)nrows) + up(sizeof(Start) * (size_t)bound) + up(8 * (size_t)nblocks) +
           up(sizeof(ChunkRec) * (size_t)bound) + up(sizeof(Shared) * (size_t)bound) + 256;
}

// Enqueues the planner's kernels on `st` (no synchronisation, nothing copied back): chunk records, cut rows and totals stay in
// the scratch `ws` (grown if needed, unless it is borrowed) at the pointers of `out`; `tables` (optional) are written by the last kernel.
// out->declined: the plan cannot be made on the device (chunk length beyond the jump table); nothing was enqueued.
hipError_t plan_chunks_device_enqueue(const int64_t *rp_dev, int64_t nrows, int64_t nz_end, int32_t S, int64_t thr, int64_t max_rows, hipStream_t st, PlanScratch *ws,
                                      DevicePlan *out, const PlanTables *tables)
{
    *out = DevicePlan();
    const int64_t cap = (int64_t)kLanes * S;
    if (thr <= 0) thr = cap / 4;
    if (thr > cap / 2) thr = cap / 2;
    if (max_rows <= 0) max_rows = INT64_MAX;
    out->thr = thr;
    out->max_rows = max_rows == INT64_MAX ? 0 : max_rows;
    if (nrows <= 0) return hipSuccess;
    if (!plan_on_device_ok(S, out->max_rows)) { out->declined = true; return hipSuccess; }
    const int64_t nblocks = (nrows + kPlanRowBlock - 1) / kPlanRowBlock, ntiles = (nrows + kTileRows - 1) / kTileRows;
    // slots <= nnz + nrows; the sum of the blocks' rooms (bound_of)
    const int64_t slots_ub = nz_end + nrows;        // (rp[0] >= 0)
    const int64_t bound = 2 * (slots_ub / cap) + (out->max_rows ? nrows / out->max_rows : 0) + 3 * nblocks;
    auto          up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t  o_tile = 0, o_q = o_tile + up(4 * (size_t)ntiles), o_j = o_q + up(4 * (size_t)(nrows + nblocks)),
                  o_st = o_j + up(2 * (size_t)nrows), o_cnt = o_st + up(sizeof(Start) * (size_t)bound), o_ch = o_cnt + up(8 * (size_t)nblocks),
                  o_sh = o_ch + up(sizeof(ChunkRec) * (size_t)bound), o_tot = o_sh + up(sizeof(Shared) * (size_t)bound), total = o_tot + 256;
    hipError_t e = hipSuccess;
    if (ws->dev_bytes < total) {
        if (ws->borrowed) return hipErrorInvalidValue;          // (an interior pointer of the caller's arena, sized by plan_scratch_bytes: the same formula as `total`)
        if (ws->dev) (void)hipFree(ws->dev);
        ws->dev = nullptr; ws->dev_bytes = 0;
        e = hipMalloc(&ws->dev, total + total / 4);
        if (e != hipSuccess) return e;
        ws->dev_bytes = total + total / 4;
    }
    uint8_t *arena = ws->dev;
    PlanArgs a;
    a.rp = reinterpret_cast<const long long *>(rp_dev);
    a.nrows = nrows; a.nblocks = (uint32_t)nblocks; a.cap = (uint32_t)cap; a.thr = (uint32_t)thr;
    a.max_rows = out->max_rows && out->max_rows < 0xffffffffll ? (uint32_t)out->max_rows : 0xffffffffu;
    a.tile_empty = reinterpret_cast<uint32_t *>(arena + o_tile);
    a.Q = reinterpret_cast<uint32_t *>(arena + o_q);
    a.J = reinterpret_cast<uint16_t *>(arena + o_j);
    a.starts = reinterpret_cast<Start *>(arena + o_st);
    a.counts = reinterpret_cast<uint32_t *>(arena + o_cnt);
    a.chunks = reinterpret_cast<ChunkRec *>(arena + o_ch);
    a.shared = reinterpret_cast<Shared *>(arena + o_sh);
    a.totals = tables && tables->totals ? tables->totals : reinterpret_cast<unsigned long long *>(arena + o_tot);
    a.odesc = tables ? tables->desc : nullptr; a.odesc2 = tables ? tables->desc2 : nullptr; a.opad = tables ? tables->pad : nullptr;
    a.onzb = tables ? reinterpret_cast<long long *>(tables->nzb) : nullptr;
    a.oroom = tables ? tables->room : 0u; a.ophased = tables && tables->phased ? 1u : 0u;
    a.nz_end = nz_end;
    if (!(tables && tables->totals)) e = hipMemsetAsync(a.totals, 0, 32, st);      // (a caller's block is zero already)
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tile_kernel, dim3((uint32_t)ntiles), dim3(kTileRows), 0, st, a);
        hipLaunchKernelGGL(q_kernel, dim3((uint32_t)ntiles), dim3(kTileRows), 0, st, a);
        hipLaunchKernelGGL(jump_kernel, dim3((uint32_t)((nrows + 255) / 256)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(walk_kernel, dim3((uint32_t)nblocks), dim3(256), (size_t)kPlanRowBlock * sizeof(uint16_t), st, a);
        hipLaunchKernelGGL(emit_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, a);
        e = hipGetLastError();
    }
    out->bound = bound;
    out->chunks = a.chunks; out->shared = a.shared; out->totals = a.totals;
    out->chunks_shared_adjacent = o_sh == o_ch + up(sizeof(ChunkRec) * (size_t)bound) && up(sizeof(ChunkRec) * (size_t)bound) == sizeof(ChunkRec) * (size_t)bound;
    return e;
}

// The plan of plan_chunks(nrows, rp, S, thr, max_rows), from a row_ptr in device memory.  *fallback is set (and nothing else
// done) when a row block holds 2^31 slots or more -- the caller then plans on the host.  One stream synchronisation for the
// counts, one for the records.
hipError_t plan_chunks_device(const int64_t *rp_dev, int64_t nrows, int64_t nz_end, int32_t S, int64_t thr, int64_t max_rows, Plan *out, bool *fallback,
                              hipStream_t st, PlanScratch *ws)
{
    *fallback = false;
    Plan &p = *out;
    p = Plan();
    p.S = S;
    p.nz_end = nz_end;
    // device scratch: the caller's (kept across the images of one cvr_create) or one of our own
    PlanScratch own;
    if (!ws) ws = &own;
    DevicePlan dp;
    hipError_t e = plan_chunks_device_enqueue(rp_dev, nrows, nz_end, S, thr, max_rows, st, ws, &dp, nullptr);
    p.thr = dp.thr; p.max_rows = dp.max_rows;
    if (e != hipSuccess || nrows <= 0) { if (own.dev) (void)hipFree(own.dev); return e; }
    if (dp.declined) { *fallback = true; return hipSuccess; }
    const int64_t bound = dp.bound;
    unsigned long long  totals_pageable[3] = {0, 0, 0};
    // Small plans come back in one go, through the caller's pinned buffer if there is one (a copy into pageable memory makes
    // the runtime wait for the stream and stage the bytes: ~25 us per call): counts first, records behind them.
    const size_t rec_bytes = (size_t)bound * (sizeof(ChunkRec) + sizeof(Shared));
    const bool   pinned = ws->pinned && ws->pinned_bytes >= 256 + rec_bytes;
    const bool   one_go = pinned || rec_bytes <= (512u << 10);
    unsigned long long *totals = pinned ? reinterpret_cast<unsigned long long *>(ws->pinned) : totals_pageable;
    uint8_t            *hch = nullptr, *hsh = nullptr;
    if (e == hipSuccess && one_go) {
        if (pinned) {
            hch = ws->pinned + 256; hsh = hch + sizeof(ChunkRec) * (size_t)bound;
        } else {
            p.chunks.resize((size_t)bound); p.shared.resize((size_t)bound);
            hch = reinterpret_cast<uint8_t *>(p.chunks.data()); hsh = reinterpret_cast<uint8_t *>(p.shared.data());
        }
        // (chunk and cut-row records are neighbours in the arena: one copy when the destination is one buffer too)
        if (pinned && dp.chunks_shared_adjacent) {
            e = hipMemcpyAsync(hch, dp.chunks, rec_bytes, hipMemcpyDeviceToHost, st);
        } else {
            e = hipMemcpyAsync(hch, dp.chunks, sizeof(ChunkRec) * (size_t)bound, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(hsh, dp.shared, sizeof(Shared) * (size_t)bound, hipMemcpyDeviceToHost, st);
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(totals, dp.totals, 24, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && (totals[2] & 1ull)) { *fallback = true; p.chunks.clear(); p.shared.clear(); }
    else if (e == hipSuccess) {
        if ((int64_t)totals[0] > bound || (int64_t)totals[1] > bound) { if (own.dev) (void)hipFree(own.dev); return hipErrorUnknown; }      // (cannot happen: bound_of)
        const size_t nc = (size_t)totals[0], ns = (size_t)totals[1];
        if (pinned && one_go) {
            p.chunks.resize(nc); p.shared.resize(ns);
            if (nc) memcpy(static_cast<void *>(p.chunks.data()), hch, sizeof(ChunkRec) * nc);
            if (ns) memcpy(static_cast<void *>(p.shared.data()), hsh, sizeof(Shared) * ns);
        } else {
            p.chunks.resize(nc); p.shared.resize(ns);
        }
        if (!one_go) {
            if (nc) e = hipMemcpyAsync(p.chunks.data(), dp.chunks, sizeof(ChunkRec) * nc, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && ns) e = hipMemcpyAsync(p.shared.data(), dp.shared, sizeof(Shared) * ns, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
    }
    if (own.dev) (void)hipFree(own.dev);
    return e;
}

void free_plan_scratch(PlanScratch &ws)
{
    if (ws.dev && !ws.borrowed) (void)hipFree(ws.dev);
    if (ws.pinned) (void)hipHostFree(ws.pinned);
    ws = PlanScratch();
}

// the longest row of a device row_ptr
hipError_t max_row_device(const int64_t *rp_dev, int64_t nrows, int64_t *out, hipStream_t st)
{
    *out = 0;
    if (nrows <= 0) return hipSuccess;
    const uint32_t      grid = (uint32_t)std::min<int64_t>(256, (nrows + 255) / 256);
    unsigned long long *d = nullptr, host[256];
    hipError_t          e = hipMalloc(&d, sizeof(host));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(max_row_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const long long *>(rp_dev), (long long)nrows, d);
    e = hipMemcpyAsync(host, d, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    for (uint32_t i = 0; i < grid; i++) *out = std::max<int64_t>(*out, (int64_t)host[i]);
    return hipSuccess;
}

// dst[i] = src[i] - base (the row pointers of a column panel, made panel-local)
hipError_t launch_shift_rows(const int64_t *src, int64_t n, int64_t base, int64_t *dst, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(shift_rows_kernel, dim3((uint32_t)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const long long *>(src), (long long)n, (long long)base, reinterpret_cast<long long *>(dst));
    return hipGetLastError();
}

hipError_t launch_block_off(const uint32_t *rows, uint32_t n, uint32_t nblocks, uint32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(block_off_kernel, dim3((nblocks + 1 + 255) / 256), dim3(256), 0, st, rows, n, nblocks, out);
    return hipGetLastError();
}

// (cvr_create's warm-up thread: asking for a kernel's attributes makes the runtime load this file's code object, which the first launch would
// otherwise wait for)
void touch_plan_kernels()
{
    hipFuncAttributes a;
    for (const void *k : {reinterpret_cast<const void *>(&tile_kernel), reinterpret_cast<const void *>(&q_kernel), reinterpret_cast<const void *>(&jump_kernel),
                          reinterpret_cast<const void *>(&walk_kernel), reinterpret_cast<const void *>(&emit_kernel)})
        (void)hipFuncGetAttributes(&a, k);
    (void)hipGetLastError();
}

}  // namespace cvr

// cvr_plan.cpp -- host planner: cuts the slot stream of a CSR matrix into chunks of exactly 64*S slots.
//
// Counterpart of the reference's per-thread partition (nnz-balanced contiguous ranges, spmv.cpp:584-627,
// and the first/last-row binary searches, spmv.cpp:631-667), re-derived for thousands of wavefront-sized
// chunks: the reference cuts every range at an arbitrary nnz (so the first and last row of every thread
// need `#pragma omp atomic`, spmv.cpp:1280-1282, 1640-1649); here a chunk ends at a row boundary whenever
// the next row is short, and only rows longer than `thr` are cut.  What is left of the chunk is a pad
// segment of zeros (the reference pads nnz to a multiple of 16 instead, spmv.cpp:474-482).
#include "cvr_plan.h"

#include <algorithm>
#include <cstdint>

namespace cvr {

int64_t plan_bound(int64_t nrows, int64_t nnz, int32_t S)
{
    // every chunk but possibly the last is > 3/4 full when thr <= cap/4; generous bound for any thr:
    // a chunk always takes at least one slot, and at least min(cap, thr+1)... keep it simple and safe.
    const int64_t cap = (int64_t)kLanes * S;
    const int64_t slots = nnz + nrows;  // upper bound of slots
    return 2 * (slots / cap + 1) + 2;
}

Plan plan_chunks(int64_t nrows, const int64_t *rp, int32_t S, int64_t thr, int64_t max_rows)
{
    Plan p;
    p.S = S;
    const int64_t cap = (int64_t)kLanes * S;
    if (thr <= 0) thr = cap / 4;
    if (thr > cap / 2) thr = cap / 2;      // keeps every chunk at least half full (plan_bound relies on it)
    p.thr = thr;
    if (max_rows <= 0) max_rows = INT64_MAX;      // column phases: a chunk's rows are accumulated in LDS, so their number is capped
    p.max_rows = max_rows == INT64_MAX ? 0 : max_rows;
    int64_t r = 0, off = 0;
    while (r < nrows) {
        Chunk c;
        c.row_first = r;
        c.nz_begin = rp[r] + off;
        c.head_shared = off > 0;
        int64_t used = 0;
        while (r < nrows) {
            // fast path: eight whole rows at a time while they fit (the slot counts are summed as a tree, so the loop
            // carries one addition per eight rows instead of one add, compare and branch per row); same result as
            // taking them one by one: no prefix of the eight can fill the chunk exactly unless all eight do
            if (off == 0) {
                bool full = false;
                while (r + 8 <= nrows && r + 8 - c.row_first <= max_rows) {
                    const int64_t *q = rp + r;
                    const int64_t  d0 = q[1] - q[0], d1 = q[2] - q[1], d2 = q[3] - q[2], d3 = q[4] - q[3];
                    const int64_t  d4 = q[5] - q[4], d5 = q[6] - q[5], d6 = q[7] - q[6], d7 = q[8] - q[7];
                    const int64_t  s8 = ((d0 > 0 ? d0 : 1) + (d1 > 0 ? d1 : 1)) + ((d2 > 0 ? d2 : 1) + (d3 > 0 ? d3 : 1)) +
                                       (((d4 > 0 ? d4 : 1) + (d5 > 0 ? d5 : 1)) + ((d6 > 0 ? d6 : 1) + (d7 > 0 ? d7 : 1)));
                    if (used + s8 > cap) break;
                    used += s8; r += 8;
                    if (used == cap) { full = true; break; }
                }
                if (full || r >= nrows) break;
            }
            if (r - c.row_first >= max_rows) break;        // row cap reached: the rest of the chunk is padding
            const int64_t len = rp[r + 1] - rp[r] - off;   // what is left of row r (off > 0 implies len > 0)
            const int64_t slots = len > 0 ? len : 1;        // an empty row owns one pad slot
            if (used + slots <= cap) {
                used += slots; r++; off = 0;
                if (used == cap) break;
                continue;
            }
            if (slots > thr) { off += cap - used; used = cap; }   // long row: cut it, fill the chunk
            break;
        }
        c.tail_shared = off > 0;
        const int64_t last = c.tail_shared ? r : r - 1;
        c.nrows_in = last - c.row_first + 1;
        c.pad_cnt = cap - used;
        c.nseg = c.nrows_in + (c.pad_cnt > 0 ? 1 : 0);
        const int64_t k = (int64_t)p.chunks.size();
        if (c.head_shared) {
            const bool ends_here = !(c.tail_shared && last == c.row_first);
            if (ends_here) p.shared.back().c1 = k;
        }
        if (c.tail_shared && !(c.head_shared && last == c.row_first)) p.shared.push_back({last, k, -1});
        p.chunks.push_back(c);
    }
    p.nz_end = nrows > 0 ? rp[nrows] : 0;
    return p;
}

}  // namespace cvr

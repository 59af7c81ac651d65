// cvr_plan.h -- host planner interface (see cvr_plan.cpp)
#pragma once
#include <cstdint>
#include <vector>

#include "cvr_format.h"

namespace cvr {

struct Chunk {
    int64_t nz_begin;     // first CSR element of the chunk (the next chunk's nz_begin is its end)
    int64_t row_first;    // first row with a segment here
    int64_t nrows_in;     // rows row_first .. row_first + nrows_in - 1 have a segment here
    int64_t nseg;         // nrows_in + (pad_cnt > 0)
    int64_t pad_cnt;      // slots of the trailing pad segment
    bool    head_shared;  // row_first began in an earlier chunk
    bool    tail_shared;  // the last row continues in the next chunk
};

struct Plan {
    int32_t S = 0;
    int64_t thr = 0;
    int64_t max_rows = 0;   // 0 = no cap
    int64_t nz_end = 0;
    std::vector<Chunk>  chunks;
    std::vector<Shared> shared;
};

int64_t plan_bound(int64_t nrows, int64_t nnz, int32_t S);
// nthreads: host threads for the row blocks (0 = up to 8; 1 when the caller already runs one planner per thread)
Plan    plan_chunks(int64_t nrows, const int64_t *row_ptr, int32_t S, int64_t split_threshold, int64_t max_rows = 0, int nthreads = 0);

}  // namespace cvr

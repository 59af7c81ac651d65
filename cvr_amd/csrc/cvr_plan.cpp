// cvr_plan.cpp -- host planner: cuts the slot stream of a CSR matrix into chunks of exactly 64*S slots.
//
// Counterpart of the reference's per-thread partition (nnz-balanced contiguous ranges, spmv.cpp:584-627,
// and the first/last-row binary searches, spmv.cpp:631-667), re-derived for thousands of wavefront-sized
// chunks: the reference cuts every range at an arbitrary nnz (so the first and last row of every thread
// need `#pragma omp atomic`, spmv.cpp:1280-1282, 1640-1649); here a chunk ends at a row boundary whenever
// the next row is short, and only rows longer than `thr` are cut.  What is left of the chunk is a pad
// segment of zeros (the reference pads nnz to a multiple of 16 instead, spmv.cpp:474-482).
#include "cvr_plan.h"

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>

namespace cvr {

int64_t plan_bound(int64_t nrows, int64_t nnz, int32_t S)
{
    // every chunk but possibly the last is > 3/4 full when thr <= cap/4; generous bound for any thr:
    // a chunk always takes at least one slot, and at least min(cap, thr+1)... keep it simple and safe.
    const int64_t cap = (int64_t)kLanes * S;
    const int64_t slots = nnz + nrows;  // upper bound of slots
    return 2 * (slots / cap + 1) + 2 + nrows / kPlanRowBlock + 1;      // + the padded chunk before every row-block restart
}

namespace {

// A few persistent worker threads for the row blocks of the planner.  (An OpenMP region would do, but the first one of a
// process initialises the OpenMP runtime -- ~100 ms on a 128-core host -- and spawning threads per call costs as much as
// the whole walk of a web-Google-sized matrix.)  Created on first use, joined at process exit.
class BlockPool {
public:
    static BlockPool &get() { static BlockPool p; return p; }
    // runs fn(0 .. n-1) on up to `team` threads including the caller; one batch at a time
    void run(int64_t n, int team, const std::function<void(int64_t)> &fn)
    {
        // After fork() the child inherits this object but none of its threads (and possibly a mutex locked by a thread that no
        // longer exists): a child runs the blocks on the calling thread and never touches the pool's state.
        if (getpid() != owner_) { for (int64_t i = 0; i < n; i++) fn(i); return; }
        std::lock_guard<std::mutex> batch(batch_mu_);
        if (team > (int)workers_.size() + 1) team = (int)workers_.size() + 1;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); want_ = team - 1; running_ = 0; gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return running_ == 0 && want_ <= 0; });
        fn_ = nullptr;
    }

private:
    BlockPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        const int nw = (int)std::min<unsigned>(hw > 1 ? hw - 1 : 0, 7);
        for (int i = 0; i < nw; i++) workers_.emplace_back([this] { loop(); });
    }
    ~BlockPool()
    {
        if (getpid() != owner_) { for (auto &t : workers_) t.detach(); return; }      // (a forked child: there is nothing to join)
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void work()
    {
        for (;;) {
            const int64_t i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || (gen_ != seen && want_ > 0); });
            if (stop_) return;
            seen = gen_;
            want_--; running_++;
            lk.unlock();
            work();
            lk.lock();
            running_--;
            if (running_ == 0 && want_ <= 0) done_.notify_all();
        }
    }
    const pid_t              owner_ = getpid();      // the process that created the threads
    std::vector<std::thread> workers_;
    std::mutex               mu_, batch_mu_;
    std::condition_variable  cv_, done_;
    const std::function<void(int64_t)> *fn_ = nullptr;
    int64_t                  n_ = 0;
    std::atomic<int64_t>     next_{0};
    int                      want_ = 0, running_ = 0;
    uint64_t                 gen_ = 0;
    bool                     stop_ = false;
};

}  // namespace

// the greedy walk over rows [r_lo, r_hi): chunks and the rows cut over them (chunk numbers local to the range)
static void plan_range(int64_t r_lo, int64_t r_hi, const int64_t *rp, int64_t cap, int64_t thr, int64_t max_rows,
                       std::vector<Chunk> &chunks, std::vector<Shared> &shared)
{
    int64_t r = r_lo, off = 0;
    const int64_t nrows = r_hi;
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
        const int64_t k = (int64_t)chunks.size();
        if (c.head_shared) {
            const bool ends_here = !(c.tail_shared && last == c.row_first);
            if (ends_here) shared.back().c1 = k;
        }
        if (c.tail_shared && !(c.head_shared && last == c.row_first)) shared.push_back({last, k, -1});
        chunks.push_back(c);
    }
}

// The walk restarts at every multiple of kPlanRowBlock rows (the chunk before it is padded): the blocks are independent, so
// they are planned by a few persistent host threads (the reference partitions per thread in parallel too, spmv.cpp:584-667); the plan does
// not depend on the number of threads.  Cost: half a chunk of padding per block on average.
Plan plan_chunks(int64_t nrows, const int64_t *rp, int32_t S, int64_t thr, int64_t max_rows, int nthreads)
{
    Plan p;
    p.S = S;
    const int64_t cap = (int64_t)kLanes * S;
    if (thr <= 0) thr = cap / 4;
    if (thr > cap / 2) thr = cap / 2;      // keeps every chunk at least half full unless a row block or the row cap ends it
    p.thr = thr;
    if (max_rows <= 0) max_rows = INT64_MAX;      // column phases: a chunk's rows are accumulated in LDS, so their number is capped
    p.max_rows = max_rows == INT64_MAX ? 0 : max_rows;
    const int64_t nblocks = (nrows + kPlanRowBlock - 1) / kPlanRowBlock;
    if (nblocks <= 1) {
        plan_range(0, nrows, rp, cap, thr, max_rows, p.chunks, p.shared);
    } else {
        std::vector<std::vector<Chunk>>  bc((size_t)nblocks);
        std::vector<std::vector<Shared>> bs((size_t)nblocks);
        int team = nthreads > 0 ? nthreads : 8;
        if ((int64_t)team > nblocks) team = (int)nblocks;
        auto block = [&](int64_t b) { plan_range(b * kPlanRowBlock, std::min(nrows, (b + 1) * kPlanRowBlock), rp, cap, thr, max_rows, bc[(size_t)b], bs[(size_t)b]); };
        if (team > 1) BlockPool::get().run(nblocks, team, block);
        else for (int64_t b = 0; b < nblocks; b++) block(b);
        size_t nc = 0, ns = 0;
        for (int64_t b = 0; b < nblocks; b++) { nc += bc[(size_t)b].size(); ns += bs[(size_t)b].size(); }
        p.chunks.reserve(nc);
        p.shared.reserve(ns);
        for (int64_t b = 0; b < nblocks; b++) {
            const int64_t k0 = (int64_t)p.chunks.size();
            p.chunks.insert(p.chunks.end(), bc[(size_t)b].begin(), bc[(size_t)b].end());
            for (Shared sh : bs[(size_t)b]) { sh.c0 += k0; sh.c1 += k0; p.shared.push_back(sh); }
        }
    }
    p.nz_end = nrows > 0 ? rp[nrows] : 0;
    return p;
}

}  // namespace cvr

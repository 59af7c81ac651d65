// cvr_layout.hip -- the layout of one CVR64 image and its device side: chunk length (pick_steps), the chunk plan with the LDS budget of
// the SpMV workgroup (plan_part), the automatic choice of the workgroup layout from a device-side look at the CSR (auto_layout), hub
// tables (choose_hubs), allocation and upload (build_part, finish_part).  Called by cvr_create (cvr_capi.hip).
#include "cvr_internal.h"

using namespace cvrh;

namespace cvrh {

int pick_steps(int64_t nslots_est, int64_t max_row, double cus, int64_t nrows)
{
    // The plain layout (one chunk per workgroup).  Images of more than 12 chunks per CU at S = 32 run in rounds and take
    // S = 32 (LiveJournal panels, R-MAT, banded: within 1 % of the best S, profiles/r02_steps_rule_check.log,
    // r01_steps_large_matrices.log).  Smaller ones are resident at once: what decides there is (1) that no row is cut over
    // chunks -- a cut row brings the fix-up kernel, a second launch worth 1.9 us on a 8-us SpMV -- so 16 S >= the longest
    // row, and (2) beyond that as many chunks as possible, i.e. the smallest such S (web-Google-shaped matrices of 0.6 M and
    // 1.3 M non-zeros: S = 28 is the best of 8 .. 64, 7.9 and 9.5 us; the round-1 fit on shards of one matrix took 24 and 44:
    // 9.8 and 11.8 us).  cvr_tune measures instead.
    const double kCus = cus;             // (a column panel that runs on one XCD counts its chunks against that XCD's 32 CUs)
    auto chunks = [&](int S) { return (double)nslots_est * 1.004 / (64.0 * S) + 1.0; };
    // (rows of fewer than four slots on average -- the road-network-like hold-out shape, 2.6 per row -- : a chunk of 32 steps holds ~800 rows, whose sums leave through
    // one wavefront's staged write-out; half the length runs 6-7 % faster: 43.0 against 46.0 us, profiles/r06_holdout.log)
    if (chunks(32) > kCus * 12.0) return nrows > 0 && nslots_est < 4 * nrows ? 16 : 32;
    int S = (int)std::min<int64_t>(64, std::max<int64_t>(12, ((max_row + 15) / 16 + 3) / 4 * 4));
    while (S < 64 && chunks(S) > kCus * 12.0) S += 4;
    return S;
}

// The workgroup shape and LDS budget of an image with chunk length pp.S (set by the caller): wavefronts per workgroup, column phases,
// x window, row-tag width, row accumulators.  Returns the cap on the rows of a chunk the planner has to keep (0 = none).
int64_t plan_layout(PartPlan &pp, int64_t ncols, bool f32, const IOpt &opt)
{
    // Wavefronts (consecutive chunks) per SpMV workgroup: 1 by default; more only pay together with an LDS window of x,
    // which the workgroup's chunks then share (profiles/r02_wg_window_sweep.log).
    pp.wpb = std::min(std::max(opt.waves_per_block, 1), cvr::kMaxWavesPerBlock);
    pp.phases = std::min(std::max(opt.col_phases, 1), 64);
    if (ncols < 64 * pp.phases) pp.phases = 1;
    if (pp.hub_n > 0) pp.phases = 1;               // (the hub flag and the row field of a phased image share bits of the column word)
    // interleaved chunks: planned and run like an image with column phases (every slot carries its row, the rows' sums live in LDS);
    // four chunks per workgroup unless the caller says otherwise, no window
    pp.ilv = opt.interleave > 0 && pp.hub_n == 0;
    if (pp.ilv) { pp.phases = 2; pp.wpb = opt.waves_per_block <= 0 ? 4 : std::min(pp.wpb, 8); }      // (spmv_ilv_kernel: at most eight wavefronts)
    // LDS window of x per workgroup (off by default): `win` consecutive values of x staged with coalesced loads; gathers
    // inside it are served by ds_read instead of a 128-byte L1 fill each.
    pp.win = std::min<int64_t>(opt.x_window < 0 ? 0 : opt.x_window, ncols + 1) & ~(int64_t)3;      // whole 16-byte loads, inside x_ext
    if (pp.ilv) pp.win = 0;
    // gang chunks: the workgroup's chunks sorted together and walked by its wavefronts in turn (spmv_gang_kernel); at least two of them
    pp.gang = pp.ilv && opt.gang > 0 && pp.wpb >= 2;
    const int64_t vs = f32 ? 4 : 8;
    int64_t       max_rows = 0;
    if (pp.phases > 1) {
        // column phases: every chunk accumulates its rows in LDS, so the planner caps the rows of a chunk at what is left of
        // the 160 KiB beside steal slots, dictionary and window -- and at what the row field of a segment's last column
        // word can hold (the bits between the column index and the end flag)
        pp.col_bits = 1;
        while (((int64_t)1 << pp.col_bits) <= std::max<int64_t>(ncols, opt.col_span)) pp.col_bits++;      // (column panels of one launch: one width for all)
        // (an interleaved column word has no end flag -- every slot ends a piece --, so its row field has one bit more; and its chunks may
        // take every accumulator the LDS holds: 5 052 fp64 rows for each of four wavefronts)
        // (gang chunks: the column word holds an offset of kGangOffBits bits from the group's first column, whatever the image's width, and a tag of
        // kGangTagBits bits = chunk inside the gang * accumulators + row: a chunk's share of the tags is its row field)
        if (pp.gang) pp.col_bits = cvr::kGangOffBits;
        const int64_t row_field = pp.gang ? (((int64_t)1 << cvr::kGangTagBits) / pp.wpb) - 1 : pp.ilv ? (pp.col_bits < 32 ? ((int64_t)1 << (32 - pp.col_bits)) - 1 : 0) : pp.col_bits < 31 ? ((int64_t)1 << (31 - pp.col_bits)) - 1 : 0;
        auto rows_for = [&](int64_t win) {
            const int64_t left = (int64_t)cvr::kLdsBytes - (cvr::kDictMax + win + 8) * vs;      // (no steal slots: spmv_seg_kernel; window + zero slot + the epilogue's arrival counter)
            int64_t rows = std::min<int64_t>((left / pp.wpb / vs) & ~(int64_t)3, pp.ilv ? 2 * (int64_t)cvr::kYStageMax : cvr::kYStageMax);
            if (const char *e = cvr::debug_env("ilv_rows_cap")) if (pp.ilv) rows = std::min<int64_t>(rows, std::max<int64_t>(64, atoll(e) & ~(int64_t)3));      // (experiments: fewer accumulators per chunk -> several workgroups per CU)
            return rows;
        };
        while (pp.win > 0 && rows_for(pp.win) < 512) pp.win = (pp.win - 1024 > 0 ? pp.win - 1024 : 0) & ~(int64_t)3;      // the window gives way
        // a chunk of S steps holds at most 64 S rows: no need for more accumulators than that (keeps the LDS small)
        const int64_t want = std::min<int64_t>(rows_for(pp.win), ((int64_t)cvr::kLanes * pp.S + 1 + 3) & ~(int64_t)3);
        // wide row tags (16 bits of their own per slot) when the column word has no room for the rows such a chunk may hold
        pp.tag16 = opt.row_tags16 > 0 || (opt.row_tags16 < 0 && row_field + 1 < want) || (pp.ilv && row_field + 1 < 64);      // (an interleaved image always has its accumulators: tags when the column word has no room at all)
        if (pp.tag16) pp.col_bits = 31;
        pp.stage = std::min<int64_t>(want, pp.tag16 ? (pp.gang ? ((int64_t)65536 / pp.wpb) & ~(int64_t)3 : (int64_t)65532) : (row_field + 1) & ~(int64_t)3);
        if (pp.stage < 64) { pp.lds_short = true; pp.phases = 1; pp.stage = 64; }
        else max_rows = pp.stage - 1;                 // + the dump entry of the pad segment
    }
    return max_rows;
}

// LDS budget without phases, once the plan's fullest chunk is known (pp.max_nseg): steal slots and dictionary are fixed; the row-sum
// stage is sized for the chunk with the most segments, so every chunk writes its y coalesced (chunks of very short rows beyond the
// stage store directly); the window takes what it asked for, the stage at least 64 rows per wavefront, and whatever does not fit is
// cut: first the stage down to 512 rows per wavefront, then the window.
void plan_stage(PartPlan &pp, bool f32)
{
    if (pp.phases == 1) {
        const int64_t vs = f32 ? 4 : 8, total = (int64_t)cvr::kLdsBytes / vs;
        const int64_t fixed = (int64_t)pp.wpb * cvr::kLanes + cvr::kDictMax + 4 + ((pp.hub_n + 3) & ~(int64_t)3);     // dictionary room is reserved before it is known
        int64_t stage = std::min<int64_t>(std::max<int64_t>((pp.max_nseg + 63) / 64 * 64, 64), cvr::kYStageMax);
        if (const char *cap = cvr::debug_env("ystage_cap")) stage = std::min<int64_t>(stage, std::max<int64_t>(64, atoll(cap) & ~(int64_t)63));      // (experiments: occupancy against staged write-out)
        if (fixed + pp.wpb * stage + pp.win > total) stage = std::max<int64_t>(std::min<int64_t>(stage, 512), ((total - fixed - pp.win) / pp.wpb) & ~(int64_t)63);
        if (stage < 64) stage = 64;
        if (fixed + pp.wpb * stage + pp.win > total) pp.win = std::max<int64_t>(0, total - fixed - pp.wpb * stage) & ~(int64_t)3;
        pp.stage = stage;
    }
}

// Chunk length of an interleaved image: as long as the row accumulators of a wavefront allow (the more non-zeros are sorted together,
// the more lanes share lines of x: profiles/r04_request_model.log) -- the rows a quarter of the LDS holds times the mean row, at least
// 16 steps, below the device planner's limit; a small matrix still gets a chunk for every wavefront of the chip.  (The panels of one
// matrix share the length: they run side by side in one launch.)
int interleave_steps(int64_t nnz, int64_t nrows, bool f32, const IOpt &opt)
{
    const int64_t vs = f32 ? 4 : 8, wpb = opt.waves_per_block > 0 ? opt.waves_per_block : 4;
    const int64_t rows = std::min<int64_t>(((int64_t)cvr::kLdsBytes / vs - cvr::kDictMax - 8) / wpb, 2 * (int64_t)cvr::kYStageMax) - 1;
    const double  mean = (double)nnz / (double)std::max<int64_t>(nrows, 1);
    const double  cus = opt.panel_on_one_xcd ? (double)opt.cus / opt.xcds : (double)opt.cus;
    int64_t       S = (int64_t)(0.85 * mean * (double)rows / 64.0) + 1;      // (a little under what the row cap fills: most chunks then end at their slots, not at their rows -- 384 / 320 steps: 343 / 332 us on the soc-LiveJournal1 shape)
    S = std::min<int64_t>(S, (int64_t)((double)nnz / (64.0 * cus * (double)wpb)) + 1);
    return (int)std::min<int64_t>(cvr::kIlvMaxSteps, std::max<int64_t>(16, (S + 3) / 4 * 4));
}

// Chunk length of interleaved column panels that run one per XCD (`rounds` panels after each other on an XCD's CUs): the launch takes as
// many GENERATIONS of workgroups as ceil(workgroups of an XCD / its CUs), every workgroup holds the LDS of a CU, so the time is
// generations x (time of a workgroup ~ S + a constant) -- soc-LiveJournal1 shape, 16 panels: 3 051 chunks (S = 416, 96 workgroups per XCD,
// three generations) 305 us, 3 074 chunks (S = 412: 97, four generations) 345 us (profiles/r04_layout_probes.log).  The chunks of a panel
// are estimated from its rows and non-zeros (chunks end at the row cap or at their slots: the cubic mean of the two counts is within 2 %
// of the planner's; 2 % are added); the result is the smallest S that needs no more generations than S = 508 does.  cvr_create checks the plan against
// it and plans once more with longer chunks when the plan has a generation more (panel_generations).
// generations of workgroups of the fullest XCD for `chunks[p]` chunks per panel, `rounds` panels per XCD dealt as cvr_create deals them
int panel_generations(const std::vector<int64_t> &chunks, int rounds, int wpb, int cus_per_xcd, double *fullest)
{
    std::vector<double> w(chunks.size());
    for (size_t p = 0; p < chunks.size(); p++) w[p] = std::ceil((double)chunks[p] / (double)wpb);
    std::sort(w.begin(), w.end(), [](double x, double y) { return x > y; });
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int    cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t p = 0; p < w.size(); p++) {
        int x = -1;
        for (int i = 0; i < 8; i++) if (cnt[i] < rounds && (x < 0 || load[i] < load[x])) x = i;
        if (x < 0) x = 0;
        load[x] += w[p]; cnt[x]++;
    }
    const double m = *std::max_element(load, load + 8);
    if (fullest) *fullest = m;
    return (int)std::ceil(m / (double)cus_per_xcd);
}

int interleave_steps_panels(const std::vector<int64_t> &nnz, const std::vector<int64_t> &nsub, int64_t col_span, int rounds, bool f32, const IOpt &opt, int *generations)
{
    const int64_t vs = f32 ? 4 : 8, wpb = opt.waves_per_block > 0 ? opt.waves_per_block : 4;
    int           cb = 1;
    while (((int64_t)1 << cb) <= col_span) cb++;
    const int64_t field = cb < 32 ? ((int64_t)1 << (32 - cb)) : 0;
    int64_t       rows = std::max<int64_t>(63, std::min<int64_t>(std::min<int64_t>((((int64_t)cvr::kLdsBytes / vs - cvr::kDictMax - 8) / wpb) & ~(int64_t)3, 2 * (int64_t)cvr::kYStageMax), field & ~(int64_t)3) - 1);
    if (const char *e = cvr::debug_env("ilv_rows_cap")) rows = std::min<int64_t>(rows, std::max<int64_t>(63, (atoll(e) & ~(int64_t)3) - 1));
    const double  cus = (double)opt.cus / opt.xcds * (cvr::debug_env("ilv_wgs_per_cu") ? atof(cvr::debug_env("ilv_wgs_per_cu")) : 1.0);      // (experiments: several workgroups per CU fill a generation)
    const size_t  P = nnz.size();
    const double margin = cvr::debug_env("ilv_est_percent") ? atof(cvr::debug_env("ilv_est_percent")) / 100.0 : 1.02;      // (a test hook: an estimate that is too low makes cvr_create plan twice)
    auto workgroups = [&](int64_t S) {                 // of the XCD with the most
        std::vector<int64_t> c(P);
        for (size_t p = 0; p < P; p++) {
            const double a = (double)nsub[p] / (double)rows, b = (double)nnz[p] / (64.0 * (double)S);
            c[p] = (int64_t)std::ceil(std::cbrt(a * a * a + b * b * b) * margin);
        }
        double m = 0;
        (void)panel_generations(c, rounds, (int)wpb, (int)cus, &m);
        return m;
    };
    // the fewest generations the longest chunks allow, then the shortest chunks that still make it (the time of a generation depends
    // little on S: soc-LiveJournal1 shape three generations 305-311 us at S = 416 .. 508, four generations 323-351 us at S = 284 .. 412)
    const double  gmin = std::ceil(workgroups(cvr::kIlvMaxSteps) / cus);
    int64_t       best = cvr::kIlvMaxSteps;
    for (int64_t S = cvr::kIlvMaxSteps - 4; S >= 16; S -= 4) {
        if (std::ceil(workgroups(S) / cus) > gmin) break;
        best = S;
    }
    if (generations) *generations = (int)gmin;
    return (int)best;
}

hipError_t plan_part(PartPlan &pp, int64_t nrows, int64_t ncols, bool f32, const int64_t *rp, const IOpt &opt, const DevRows *dr)
{
    const int64_t nz0 = rp ? (nrows ? rp[0] : 0) : dr->nz0, nz1 = rp ? (nrows ? rp[nrows] : 0) : dr->nz1;
    pp.S = opt.steps_per_chunk;
    if (pp.S == 0 && opt.interleave > 0 && nrows > 0) pp.S = interleave_steps(nz1 - nz0, nrows, f32, opt);
    if (pp.S == 0) {
        int64_t max_row = 0;
        const double cus = opt.panel_on_one_xcd ? (double)opt.cus / opt.xcds : (double)opt.cus;
        if ((double)(nz1 - nz0 + nrows / 4) / (64.0 * 32.0) <= cus * 12.0) {    // (only where the rule weighs single launches)
            if (rp) for (int64_t r = 0; r < nrows; r++) max_row = std::max(max_row, rp[r + 1] - rp[r]);
            else { const hipError_t e = cvr::max_row_device(dr->rp, nrows, &max_row, dr->st); if (e != hipSuccess) return e; }
        }
        pp.S = pick_steps(nz1 - nz0 + nrows / 4, max_row, cus, nrows);
    }
    const int64_t max_rows = plan_layout(pp, ncols, f32, opt);
    if (rp) {
        pp.plan = cvr::plan_chunks(nrows, rp, pp.S, opt.split_threshold, max_rows, pp.plan_threads);
    } else {
        bool             declined = false;
        const hipError_t e = cvr::plan_chunks_device(dr->rp, nrows, nz1, pp.S, opt.split_threshold, max_rows, &pp.plan, &declined, dr->st, dr->ws);
        if (e != hipSuccess) return e;
        if (declined) {       // (chunks beyond the 15-bit jump, or a row block beyond 32-bit slot positions): the row pointers come to the host after all
            std::vector<int64_t> hrp((size_t)nrows + 1);
            const hipError_t     e2 = hipMemcpy(hrp.data(), dr->rp, sizeof(int64_t) * hrp.size(), hipMemcpyDeviceToHost);
            if (e2 != hipSuccess) return e2;
            pp.plan = cvr::plan_chunks(nrows, hrp.data(), pp.S, opt.split_threshold, max_rows, pp.plan_threads);
        }
    }
    const cvr::Plan &plan = pp.plan;
    const int64_t    nchunks = (int64_t)plan.chunks.size();
    pp.yext = nrows + 1 + 2 * nchunks;
    if (pp.yext >= (int64_t)0xffffffffu || nchunks >= (int64_t)0x7fffffff) { pp.too_large = true; return hipSuccess; }
    pp.desc.resize((size_t)nchunks * 4);
    if (pp.phases > 1) pp.desc2.resize((size_t)nchunks * 2, 0u);
    pp.pad.resize((size_t)nchunks);
    pp.nzb.resize((size_t)nchunks + 1);
    for (int64_t k = 0; k < nchunks; k++) {
        const cvr::Chunk &c = plan.chunks[(size_t)k];
        pp.max_nseg = std::max(pp.max_nseg, c.nseg);
        pp.desc[4 * k + 0] = (uint32_t)c.row_first;
        pp.desc[4 * k + 1] = (uint32_t)c.nseg;
        // where segment q writes: a row begun earlier -> carry_head(k); a row continued later -> carry_tail(k);
        // the pad segment -> dump; else its row.  head_dest / last_dest are that rule at q = 0 and q = nseg-1.
        auto dest = [&](int64_t q) -> uint32_t {
            if (q >= c.nrows_in) return (uint32_t)nrows;
            if (q == 0 && c.head_shared) return (uint32_t)(nrows + 1 + 2 * k);
            if (q == c.nrows_in - 1 && c.tail_shared) return (uint32_t)(nrows + 1 + 2 * k + 1);
            return (uint32_t)(c.row_first + q);
        };
        pp.desc[4 * k + 2] = dest(0);
        pp.desc[4 * k + 3] = dest(c.nseg - 1);
        if (pp.phases > 1) {            // column phases: head / last_dest belong to the first / last ROW; desc.y and desc2.x come from the device
            pp.desc[4 * k + 3] = dest(c.nrows_in - 1);
            pp.desc2[2 * k + 1] = (uint32_t)c.nrows_in;
        }
        pp.pad[(size_t)k] = (uint32_t)c.pad_cnt;
        pp.nzb[(size_t)k] = c.nz_begin;
    }
    pp.nzb[(size_t)nchunks] = plan.nz_end;
    if (pp.phases > 1) {          // no more accumulators than the fullest chunk has rows (+ the dump entry): the cap stays what no chunk exceeds
        int64_t most = 0;
        for (const cvr::Chunk &c : plan.chunks) most = std::max(most, c.nrows_in);
        pp.stage = std::min<int64_t>(pp.stage, std::max<int64_t>(64, (most + 1 + 3) & ~(int64_t)3));
    }
    plan_stage(pp, f32);
    return hipSuccess;
}

// A second stream per device, shared by all handles of the process, for the few analysis / conversion kernels that do not
// depend on each other (layout probe | dictionary scan, conversion | window choice): each of them is too small to fill the
// GPU and bound by latency, so side by side they take the time of one.  Created on first use (creating a stream costs
// milliseconds), never destroyed.
hipStream_t side_stream(int device, int which)
{
    static std::mutex  mu;
    static hipStream_t streams[64][2] = {};
    if (device < 0 || device >= 64 || which < 0 || which > 1) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    hipStream_t &st = streams[device][which];
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
    return st;
}

// The handle's own stream comes from a small per-device pool: creating a stream takes 2-3 ms and destroying one about as long,
// more than the whole analysis and conversion of a web-Google-sized matrix.  A stream goes back idle (cvr_destroy synchronises
// it first); at most eight are kept per device.
static std::mutex               g_pool_mu;
static std::vector<hipStream_t> g_stream_pool[64];

hipError_t acquire_stream(int device, hipStream_t *out)
{
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_stream_pool[device].empty()) { *out = g_stream_pool[device].back(); g_stream_pool[device].pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void release_stream(int device, hipStream_t s)
{
    if (!s) return;
    if (device >= 0 && device < 64 && hipStreamSynchronize(s) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_stream_pool[device].size() < 8) { g_stream_pool[device].push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}


// the dictionary scan of the values [nz0, nz1) of a part, enqueued on the handle's stream: table and flags come back into
// tab_host / flags_host once the stream is synchronised
hipError_t enqueue_dict_scan(cvr_handle *h, const void *d_va, int64_t nz0, int64_t nz1, bool f32, bool first, unsigned long long *tab_host, uint32_t *flags_host, bool last,
                                    hipStream_t st)
{
    unsigned long long *d_tab = reinterpret_cast<unsigned long long *>(h->d_small + kSmallDictTab);
    uint32_t           *d_flags = reinterpret_cast<uint32_t *>(h->d_small + kSmallDictFlags);
    hipError_t          e = hipSuccess;
    if (first && !h->small_clean) {        // (cvr_create left the table and the flags ready for the first scan)
        e = hipMemsetAsync(d_tab, 0xff, sizeof(unsigned long long) * 1024, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, sizeof(uint32_t) * 2, st);
    }
    if (e == hipSuccess) e = cvr::launch_dict_scan(d_va, nz0, nz1, f32, d_tab, d_flags, st);
    if (e == hipSuccess && last) {
        e = hipMemcpyAsync(tab_host, d_tab, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(flags_host, d_flags, sizeof(uint32_t) * 2, hipMemcpyDeviceToHost, st);
    }
    return e;
}

// The resident layout's candidates: 8 .. 4 chunks per workgroup, each with the smallest S that keeps the workgroups a few under the CU
// count.  The one whose chunks are closest to 46 steps wins (ties: the longer chunks, then more waves): measured layouts of the
// web-Google shape, of smaller ones and of its row shards -- whole matrix 7 x 48, half 4 x 44, a third 4 x 32 -- against the 7 x 24 or the
// plain layout that a rule of "most workgroups, at least 6 chunks each, S >= 24" gave them (profiles/r03_resident_rule.log: half the
// matrix 14.4 -> 12.8 us, a third 11.1 -> 8.9, one of two row shards 18.2 -> 14.7).
// false: tiny matrices and matrices beyond one resident pass.  CVR_RESIDENT_RULE=old keeps the former rule (diagnostics).
bool resident_candidate(double slots, int cus, int *best_w, int *best_S)
{
    *best_w = 0; *best_S = 0;
    const char *const rr = cvr::debug_env("resident_rule");          // (read per call: a latched static would keep the first call's environment)
    const bool old_rule = rr && !strcmp(rr, "old");
    if (old_rule) {
        double best_fill = 0;
        for (int w = 8; w >= 6; w--) {
            int S = (int)std::ceil(slots / (64.0 * w * (double)(cus - 4)) / 4.0) * 4;
            if (S < 24 || S > 128) continue;
            const double wgs = std::ceil(slots / (64.0 * w * S));
            if (wgs > best_fill) { best_fill = wgs; *best_w = w; *best_S = S; }
        }
        return *best_w != 0;
    }
    int best_d = 1 << 30;
    for (int w = 8; w >= 4; w--) {
        const int S = (int)std::ceil(slots / (64.0 * w * (double)(cus - 4)) / 4.0) * 4;
        if (S < 28 || S > 128) continue;              // (shorter resident chunks lose to the plain layout: 24 x 4 against S = 32 on a quarter of the matrix, 11.5 against 9.9 us)
        const int d = std::abs(S - 46);
        if (d < best_d || (d == best_d && S > *best_S)) { best_d = d; *best_w = w; *best_S = S; }
    }
    return *best_w != 0;
}

// A matrix with every layout option left to the rules that is too large for the resident layout: more slots than 8 chunks of 128 steps per
// workgroup hold, or more rows than the chunks' LDS accumulators (plan_layout's row cap) take in one pass -- web-Google shapes from ~11 M
// non-zeros on (2.2 M rows / 987 per chunk > 2 048 chunks), which the planner would find out after planning (build_part: stuck -> plain
// layout).  cvr_create asks before it decides about column panels.
bool resident_out_of_reach(int64_t nrows, int64_t nnz, int64_t ncols, bool f32, const IOpt &opt)
{
    if (opt.steps_per_chunk != 0 || opt.waves_per_block != 0 || opt.x_window >= 0 || opt.col_phases >= 0 || opt.debug_col_mask || cvr::debug_env("no_auto_layout")) return false;
    if (nrows < 4096 || ncols < 4096 || opt.cus <= 4) return false;
    const int64_t vs = f32 ? 4 : 8;
    const double  slots = ((double)nnz + (double)nrows / 4) * 1.006;
    int           w = 0, S = 0;
    if (!resident_candidate(slots, opt.cus, &w, &S)) return slots > 64.0 * 8 * 128 * (double)(opt.cus - 4);      // (else: too small for it)
    PartPlan pp;
    IOpt     spec = opt;
    spec.waves_per_block = w; spec.steps_per_chunk = S; spec.x_window = (int32_t)((96 * 1024) / vs);
    spec.col_phases = (int32_t)std::min(32.0, std::max(2.0, std::floor((double)ncols * vs / 450e3 + 0.5)));
    pp.S = S;
    const int64_t max_rows = plan_layout(pp, ncols, f32, spec);
    // (chunks close at the row cap or at their slots, whichever comes first: the web-Google shape fits up to 0.975 of the chunks by rows alone,
    // not at 0.9975 -- profiles/r03_mid_size_panels.log)
    return max_rows > 0 && (double)((nrows + max_rows - 1) / max_rows) > 0.985 * (double)w * opt.cus;
}

// The automatic layout (every layout option left at its default; one image, no column panels).  Matrices small enough
// for all their chunks to be resident at once -- 4 to 8 chunks per workgroup, one workgroup per CU -- run in the "resident"
// layout when it pays: the workgroup's chunks share an LDS window of x if a sizeable share of the non-zeros lies near the
// diagonal, and feed their rows column phase by column phase when x is larger than what an L2 keeps beside the matrix
// stream (profiles/r02_wg_window_sweep.log, r02_column_phases_sweep.log: 32.5 -> 23.4 us on the web-Google shape).
// Decided from a device-side pass over the uploaded CSR (sortedness of the rows, near-diagonal share).
int auto_layout(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, IOpt &opt, const std::function<int(const IOpt &)> *meanwhile)
{
    const int64_t nnz = nz1 - nz0;
    opt.layout_auto_resident = 0;
    if (opt.steps_per_chunk != 0 || opt.waves_per_block != 0 || opt.x_window >= 0 || opt.col_phases >= 0 || opt.debug_col_mask || opt.interleave > 0 || cvr::debug_env("no_auto_layout")) {
        if (opt.col_phases < 0) opt.col_phases = 0;
        return CVR_OK;
    }
    opt.col_phases = 0;
    if (nrows < 4096 || ncols < 4096) return CVR_OK;
    const int64_t vs = f32 ? 4 : 8;
    const double  slots = ((double)nnz + (double)nrows / 4) * 1.006;      // pad slots of empty rows, chunk tails
    int best_w = 0, best_S = 0;
    resident_candidate(slots, opt.cus, &best_w, &best_S);
    if (!best_w) return CVR_OK;
    // 96 KiB of x per workgroup: the gathers that stay outside the window are what loads the L2s (one request per gather, an XCD's
    // L2 takes ~16 per clock: profiles/r03_gather_rate_ubench.log), so the window takes what the row accumulators leave
    // (64 -> 96 KiB: 21.4 -> 20.9 us on the web-Google shape, profiles/r03_window_sizes.log)
    const int64_t win = (96 * 1024) / vs;
    static_assert(sizeof(unsigned long long) * 2 * cvr::kProbeBlocks <= kSmallDictTab, "probe output fits its part of the small scratch");
    unsigned long long *d_out = reinterpret_cast<unsigned long long *>(h->d_small + kSmallProbe);
    std::vector<unsigned long long> pageable;
    const bool          pin = h->plan_ws.pinned && h->plan_ws.pinned_bytes >= kPinnedSmall;
    if (!pin) pageable.resize(2 * cvr::kProbeBlocks + 1024 + 1);
    unsigned long long *outv = pin ? reinterpret_cast<unsigned long long *>(h->plan_ws.pinned + kPinnedProbe) : pageable.data();
    unsigned long long *tabv = pin ? reinterpret_cast<unsigned long long *>(h->plan_ws.pinned + kPinnedDictTab) : pageable.data() + 2 * cvr::kProbeBlocks;
    uint32_t           *flagv = pin ? reinterpret_cast<uint32_t *>(h->plan_ws.pinned + kPinnedDictFlags) : reinterpret_cast<uint32_t *>(pageable.data() + 2 * cvr::kProbeBlocks + 1024);
    HIP_TRY(hipStreamSynchronize(h->stream));          // the upload
    const double tp0 = now_s();
    // The probe, the dictionary scan of the values (which depends on nothing decided here) and -- `meanwhile` -- the chunk plan of
    // the layout the probe most often confirms run side by side: each is a handful of kernels too small to fill the GPU, and every
    // submission with a synchronisation of its own costs more than its kernels (the upload is complete: nothing to order between
    // the streams).  The caller keeps the speculative plan when the decision below is the layout it was made for.
    const double xbytes = (double)ncols * vs;
    const int32_t phases_rule = (int32_t)std::min(32.0, std::max(2.0, std::floor(xbytes / 450e3 + 0.5)));
    hipStream_t pstream = meanwhile ? side_stream(h->device, 1) : h->stream;
    if (!pstream) pstream = h->stream;
    hipError_t e = cvr::launch_probe(part.d_rp, part.d_ci, nrows, ncols, (uint32_t)(win / 4), d_out, pstream, h->small_clean);
    if (e == hipSuccess) e = hipMemcpyAsync(outv, d_out, sizeof(unsigned long long) * 2 * cvr::kProbeBlocks, hipMemcpyDeviceToHost, pstream);
    const bool with_dict = opt.value_dict != 0 && nnz > 0;
    hipStream_t side = with_dict ? side_stream(h->device, 0) : nullptr;
    if (!side) side = h->stream;
    if (e == hipSuccess && with_dict) e = enqueue_dict_scan(h, part.d_va, nz0, nz1, f32, true, tabv, flagv, true, side);
    double meanwhile_s = 0;
    if (e == hipSuccess && meanwhile && pstream != h->stream && side != h->stream && xbytes > 2.5e6) {
        IOpt spec = opt;
        spec.layout_auto_resident = 1; spec.waves_per_block = best_w; spec.steps_per_chunk = best_S; spec.x_window = (int32_t)win; spec.col_phases = phases_rule;
        const double tm0 = now_s();
        const int    rcm = (*meanwhile)(spec);
        meanwhile_s = now_s() - tm0;
        if (rcm) { (void)hipStreamSynchronize(pstream); (void)hipStreamSynchronize(side); return rcm; }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(pstream);
    if (e == hipSuccess && pstream != h->stream) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && side != h->stream) e = hipStreamSynchronize(side);
    unsigned long long out[2] = {0, 0};
    for (uint32_t b = 0; b < cvr::kProbeBlocks; b++) { out[0] |= outv[2 * b]; out[1] += outv[2 * b + 1]; }
    h->small_clean = false;
    if (e == hipSuccess && with_dict) { h->dict_tab.assign(tabv, tabv + 1024); h->dict_flags[0] = flagv[0]; h->dict_flags[1] = flagv[1]; h->dict_scanned = true; }
    h->info.probe_s = std::max(0.0, now_s() - tp0 - meanwhile_s);      // (the speculative plan beside it is counted as planning)
    if (e != hipSuccess) return fail(CVR_ERR_HIP, "layout probe: %s", hipGetErrorString(e));
    const bool   sorted = out[0] == 0;
    const double near = (double)out[1] / std::max<double>((double)nnz, 1.0);
    // (a matrix with nearly everything near the diagonal is a band: consecutive rows share their lines of x in L1 already)
    const bool   want_win = near >= 0.15 && near < 0.9, want_phases = sorted && xbytes > 2.5e6 && near < 0.9;
    h->info.near_diagonal_share = near;
    if (!want_win && !want_phases) return CVR_OK;
    opt.layout_auto_resident = 1;
    opt.waves_per_block = best_w;
    opt.steps_per_chunk = best_S;
    opt.x_window = want_win ? (int32_t)win : 0;
    opt.col_phases = want_phases ? phases_rule : 1;
    return CVR_OK;
}

// Hub table (cvr_hub.hip): hub_table > 0 asks for that many entries, < 0 decides: matrices too large for the resident layout
// whose x does not fit an L2 get the columns counted on the device, and the table is used when the columns that fit the LDS
// (beside 8 chunks' row stages) hold at least half of the (sampled) non-zeros -- R-MAT scale 22 fp32: 0.56, 402 -> 289 us;
// fp64 (half as many entries fit): 0.41, where the table loses (profiles/r02_hub_table_rmat.log).  The workgroup then has 8 chunks.
int choose_hubs(cvr_handle *h, Part &part, const int32_t *d_ci, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, IOpt &opt, PartPlan &pp,
                       bool allow_reorder)
{
    if (opt.hub_table == 0 || opt.layout_auto_resident || opt.col_phases > 1 || opt.interleave > 0 || nrows <= 0 || ncols >= (int64_t)cvr::kHubBit) return CVR_OK;
    const int64_t vs = f32 ? 4 : 8, nnz = nz1 - nz0;
    const bool    automatic = opt.hub_table < 0;
    if (automatic && (opt.waves_per_block != 0 || opt.x_window > 0 || opt.debug_col_mask || cvr::debug_env("no_auto_layout") || (double)ncols * vs < 6e6 || nnz < (8 << 20))) return CVR_OK;
    const int     wpb = opt.waves_per_block > 0 ? std::min(opt.waves_per_block, cvr::kMaxWavesPerBlock) : 8;
    const int64_t win = std::max(opt.x_window, 0);
    int64_t       room = ((int64_t)cvr::kLdsBytes / vs - (int64_t)wpb * (cvr::kLanes + 512) - cvr::kDictMax - win - 8) & ~(int64_t)1023;      // 512 staged row sums per chunk
    if (room < 1024) return automatic ? CVR_OK : fail(CVR_ERR_INVALID, "hub_table: no LDS left beside %d chunks per workgroup and the x window", wpb);
    if (!automatic) room = std::min<int64_t>(room, opt.hub_table);
    HIP_TRY(hipStreamSynchronize(h->stream));          // the upload
    const double t0 = now_s();
    cvr::HubSelection sel;
    // The whole of x re-ordered by popularity (every column index of the image becomes its rank, x_perm = x[perm] is built
    // before every SpMV): the popular columns then share cache lines and stay in the L2s.  R-MAT-22 fp64 (x = 33.5 MB): plain
    // 487 us, table alone 538, table + re-ordered x 400 us; fp32 (x = 16.8 MB): 288 -> 291 us, so only for a large x; not
    // inside column panels (a panel ranks its own range).  hub_reorder: < 0 = this rule, 0 off, 1 on.
    const bool        full_order = allow_reorder && (opt.hub_reorder > 0 || (opt.hub_reorder < 0 && (double)ncols * vs >= 24e6));
    const hipError_t  e = cvr::select_hubs(d_ci, nz0, nz1, ncols, (uint32_t)room, &sel, h->stream, full_order);
    h->info.hub_select_s += now_s() - t0;
    if (e != hipSuccess) { cvr::free_hubs(sel); return fail(CVR_ERR_HIP, "hub selection: %s", hipGetErrorString(e)); }
    h->info.hub_share = std::max(h->info.hub_share, sel.share);
    if (sel.H == 0 || (automatic && sel.share < (full_order ? 0.3 : 0.5))) { cvr::free_hubs(sel); return CVR_OK; }
    part.img.hub_n = sel.H; part.img.hub_cols = sel.hub_cols; part.img.hub_index = sel.hub_index; part.img.hub_bitmap = sel.hub_bitmap;
    part.img.order_n = sel.order_n;
    HIP_TRY(hipMalloc(&part.img.hub_x, (size_t)vs * (sel.order_n ? ((size_t)sel.order_n + 8) : ((sel.H + 3u) & ~3u))));
    pp.hub_n = sel.H;
    if (opt.waves_per_block == 0) opt.waves_per_block = wpb;
    return CVR_OK;
}

// the fields of the device image that follow from the layout (not the per-chunk tables): shared by build_part and the fused path
int setup_image(cvr_handle *h, Part &part, const PartPlan &pp, int64_t nrows, int64_t ncols, bool f32, int64_t nchunks, int64_t nshared, const IOpt &opt, const IOpt &popt)
{
    (void)h;
    const int S = pp.S;
    const int G = S / 4;
    cvr::DeviceImage &img = part.img;
    img.S = S; img.G = G; img.f32 = f32; img.nchunks = (uint32_t)nchunks; img.nrows = (uint32_t)nrows;
    img.pad_col = (uint32_t)ncols; img.nshared = (uint32_t)nshared;
    img.xcd_swizzle = opt.xcds != 8 ? 0 : opt.xcd_swizzle < 0 ? 1 : opt.xcd_swizzle > 6 ? 1 : opt.xcd_swizzle;
    img.ncus = (uint32_t)opt.cus;
    img.wpb = (uint32_t)pp.wpb;
    img.ystage = (uint32_t)pp.stage;
    img.phases = (uint32_t)pp.phases;
    if (pp.phases > 1) {
        const int64_t pw = ((ncols + pp.phases - 1) / pp.phases + 15) / 16 * 16;
        img.phase_width = (uint32_t)std::max<int64_t>(pw, 16);
        img.col_bits = (uint32_t)pp.col_bits;
        img.tag16 = pp.tag16;
        // pieces: a lane that sits on a long row's segment falls behind the column ranges the other lanes have moved on to; with
        // chunks longer than a few steps per phase the segments are cut (auto: 8 elements once a phase takes 8 steps or more)
        img.ilv = pp.ilv;
        img.gang = pp.gang ? (uint32_t)pp.wpb : 0u;
        if (cvr::debug_env("phase_clocks") && !pp.ilv && !f32) {          // room for the time stamps of every wavefront of the launch (spmv_seg_kernel<.., PROF>)
            const size_t words = ((size_t)nchunks / (size_t)std::max(pp.wpb, 1) + 16) * 16 * 8;
            if (hipMalloc(&img.prof, words * sizeof(unsigned long long)) == hipSuccess) { img.prof_words = (uint32_t)words; (void)hipMemset(img.prof, 0, words * sizeof(unsigned long long)); }
            else { (void)hipGetLastError(); img.prof = nullptr; }
        }
        // (a power of two, rounded down: the segment-table kernel cuts with shifts)
        img.piece_max = opt.piece_max > 0 ? 1u << (31 - __builtin_clz((uint32_t)opt.piece_max)) : opt.piece_max < 0 && S / pp.phases >= 8 && !opt.panel_on_one_xcd ? 8u : 0u;
        img.col_mask = pp.tag16 ? cvr::kColMask : (1u << pp.col_bits) - 1u;
        if (pp.ilv) img.piece_max = 1;      // (every slot is a piece of its own)
    }
    if (popt.col_phases > 1 && pp.lds_short && !popt.layout_auto_resident) return fail(CVR_ERR_INVALID, "col_phases: no room for at least 63 row accumulators per chunk (LDS beside %d waves per workgroup and the x window, or %d-bit column indices)", pp.wpb, pp.col_bits);
    const int64_t win = pp.win;
    img.win_elems = (uint32_t)win;
    if (opt.debug_col_mask) img.col_mask &= (uint32_t)opt.debug_col_mask & cvr::kColMask;   // profiling knob (tools/sweep.py --colmask)
    if (const char *e = cvr::debug_env("stream_mod")) img.stream_mod = (uint32_t)std::max(0, atoi(e));

    return CVR_OK;
}

// The chunk plans of a device split's column panels as one submission: every panel's planner is enqueued (three streams take turns,
// each with scratch of its own), its last kernel writes nzb / pad / desc of the panel's image on the device, the cut rows are copied
// beside them, and only the counts come back -- one synchronisation for all panels instead of two per panel with the plan's records in
// between (sixteen panels of the LiveJournal shape: 5.0 -> ~1 ms).  For panels without hub tables and phases whose chunk length does
// not depend on a look at the rows; *done = false: the caller plans panel by panel (nothing is left allocated).
int plan_panels_batched(cvr_handle *h, const cvr::DeviceSplit &d, const std::vector<int64_t> &nsubs, const std::vector<int64_t> &pcols, bool f32, const std::vector<IOpt> &popts,
                        std::vector<PartPlan> &pps, std::vector<DevRows> &drs, bool *done)
{
    *done = false;
    const int P = (int)h->parts.size();
    if (P < 2 || cvr::debug_env("serial_panel_plans") || cvr::debug_env("host_plan")) return CVR_OK;
    std::vector<int64_t> bound((size_t)P), maxr((size_t)P);
    size_t               scratch = 0;
    for (int p = 0; p < P; p++) {
        const IOpt   &o = popts[(size_t)p];
        const int64_t ns = nsubs[(size_t)p], nzp = d.off[p + 1] - d.off[p];
        PartPlan     &pp = pps[(size_t)p];
        if (ns <= 0 || nzp <= 0 || o.hub_table != 0) return CVR_OK;
        pp.S = o.steps_per_chunk;
        if (pp.S == 0) {
            const double cus = o.panel_on_one_xcd ? (double)o.cus / o.xcds : (double)o.cus;
            if ((double)(nzp + ns / 4) / (64.0 * 32.0) <= cus * 12.0) return CVR_OK;          // (the chunk length would depend on the longest row: plan_part)
            pp.S = pick_steps(nzp + ns / 4, 0, cus);
        }
        maxr[(size_t)p] = plan_layout(pp, pcols[(size_t)p], f32, o);
        if (!cvr::plan_on_device_ok(pp.S, maxr[(size_t)p])) return CVR_OK;
        if ((pp.phases > 1 && !pp.ilv) || pp.hub_n > 0) return CVR_OK;      // (interleaved panels: planned like phased images, their tables written on the device too)
        bound[(size_t)p] = cvr::plan_bound_device(ns, nzp, pp.S, maxr[(size_t)p]);
        if (ns + 1 + 2 * bound[(size_t)p] >= (int64_t)0xffffffffu || bound[(size_t)p] >= (int64_t)0x7fffffff) return CVR_OK;
        scratch = std::max(scratch, cvr::plan_scratch_bytes(ns, nzp, pp.S, maxr[(size_t)p]));
    }
    hipStream_t sts[3] = {h->stream, side_stream(h->device, 0), side_stream(h->device, 1)};
    int         nst = 1;
    if (sts[1] && sts[2]) nst = 3;
    struct Own {          // (the side streams' scratch and the totals in ONE allocation: every hipFree of a large buffer takes ~190 us)
        cvr::PlanScratch ws[3];
        unsigned long long *d_tot = nullptr;
        void *base = nullptr;
        ~Own() { (void)hipFree(base); }
    } own;
    if (h->plan_ws.dev_bytes < scratch) {
        if (h->plan_ws.dev) (void)hipFree(h->plan_ws.dev);
        h->plan_ws.dev = nullptr; h->plan_ws.dev_bytes = 0;
        HIP_TRY(hipMalloc(&h->plan_ws.dev, scratch));
        h->plan_ws.dev_bytes = scratch;
    }
    own.ws[0].dev = h->plan_ws.dev; own.ws[0].dev_bytes = h->plan_ws.dev_bytes;
    {
        // the side streams' scratch and the totals: slices of the handle's planner scratch when it is large enough for all three streams (it is sized
        // for the WHOLE matrix's rows; a panel's sub-rows need a fraction) -- an allocation of their own otherwise, whose hipFree at the end of
        // this function stood ~0.2 ms in the preprocessing of the soc-LiveJournal1 shape
        const size_t sc = (scratch + 255) & ~(size_t)255, tot_bytes = sizeof(unsigned long long) * 4 * (size_t)P;
        uint8_t     *side = nullptr;
        if (h->plan_ws.dev_bytes >= sc * (size_t)nst + tot_bytes) { side = h->plan_ws.dev + sc; own.ws[0].dev_bytes = sc; own.ws[0].borrowed = true; }      // (borrowed: the planner may not grow a slice)
        else { HIP_TRY(hipMalloc(&own.base, sc * (size_t)(nst - 1) + tot_bytes)); side = static_cast<uint8_t *>(own.base); }
        for (int i = 1; i < nst; i++) { own.ws[i].dev = side + sc * (size_t)(i - 1); own.ws[i].dev_bytes = scratch; own.ws[i].borrowed = true; }
        own.d_tot = reinterpret_cast<unsigned long long *>(side + sc * (size_t)(nst - 1));
        for (int p = 0; p < P; p++) HIP_TRY(hipMemsetAsync(own.d_tot + 4 * (size_t)p, 0, sizeof(unsigned long long) * 4, sts[p % nst]));      // (in front of the panel's planner on its own stream: no synchronisation)
    }
    auto abandon = [&]() {          // (nothing of the attempt stays: the caller allocates again)
        for (int i = 0; i < nst; i++) (void)hipStreamSynchronize(sts[i]);
        release_panel_plans(h);
        return CVR_OK;
    };
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    {   // two allocations for all panels, a slice each: what goes with the CSR (row pointers, chunk starts, pad counts) and what stays with the images
        // (desc, desc2, cut rows) -- ~100 hipMalloc calls and as many hipFree calls less for sixteen panels
        size_t total = 0, tables = 0;
        for (int p = 0; p < P; p++) {
            const size_t nb = std::max<size_t>((size_t)bound[(size_t)p], 1);
            total += up256(sizeof(int64_t) * ((size_t)nsubs[(size_t)p] + 1)) + up256(sizeof(int64_t) * (nb + 1)) + up256(sizeof(uint32_t) * nb);
            tables += up256(16 * nb) + up256(24 * nb) + up256(8 * nb);
        }
        if (h->panel_rp) { (void)hipFree(h->panel_rp); h->panel_rp = nullptr; }
        if (h->panel_tables) { (void)hipFree(h->panel_tables); h->panel_tables = nullptr; }
        HIP_TRY(hipMalloc(&h->panel_rp, std::max<size_t>(total, 256)));
        HIP_TRY(hipMalloc(&h->panel_tables, std::max<size_t>(tables, 256)));
    }
    size_t rp_off = 0, tab_off = 0;
    auto carve = [&](void *base, size_t &off, size_t bytes) { void *q = static_cast<uint8_t *>(base) + off; off += up256(bytes); return q; };
    for (int p = 0; p < P; p++) {
        Part           &part = h->parts[(size_t)p];
        const int64_t   ns = nsubs[(size_t)p], nzp = d.off[p + 1] - d.off[p], nb = bound[(size_t)p];
        const hipStream_t st = sts[p % nst];
        const size_t nb1 = std::max<size_t>((size_t)nb, 1);
        part.d_rp = static_cast<int64_t *>(carve(h->panel_rp, rp_off, sizeof(int64_t) * ((size_t)ns + 1)));
        part.d_nzb = static_cast<int64_t *>(carve(h->panel_rp, rp_off, sizeof(int64_t) * (nb1 + 1)));
        part.d_pad = static_cast<uint32_t *>(carve(h->panel_rp, rp_off, sizeof(uint32_t) * nb1));
        part.rp_borrowed = true;
        HIP_TRY(cvr::launch_shift_rows(d.rp + d.sub0[p], ns + 1, d.off[p], part.d_rp, st));
        const bool phased = pps[(size_t)p].phases > 1;
        part.img.desc = static_cast<decltype(part.img.desc)>(carve(h->panel_tables, tab_off, 16 * nb1));
        part.img.shared = static_cast<decltype(part.img.shared)>(carve(h->panel_tables, tab_off, 24 * nb1));
        part.img.desc2 = phased ? static_cast<decltype(part.img.desc2)>(carve(h->panel_tables, tab_off, 8 * nb1)) : nullptr;
        part.tables_borrowed = true;
        cvr::PlanTables tables;
        tables.desc = part.img.desc; tables.desc2 = part.img.desc2; tables.pad = part.d_pad; tables.nzb = part.d_nzb; tables.room = (uint32_t)nb; tables.phased = phased;
        tables.totals = own.d_tot + 4 * (size_t)p;
        cvr::DevicePlan dp;
        HIP_TRY(cvr::plan_chunks_device_enqueue(part.d_rp, ns, nzp, pps[(size_t)p].S, popts[(size_t)p].split_threshold, maxr[(size_t)p], st, &own.ws[p % nst], &dp, &tables));
        if (dp.declined || dp.bound != nb) return abandon();
        if (nb > 0) HIP_TRY(hipMemcpyAsync(part.img.shared, dp.shared, 24 * (size_t)nb, hipMemcpyDeviceToDevice, st));      // (the scratch goes to the stream's next panel)
    }
    for (int i = 0; i < nst; i++) HIP_TRY(hipStreamSynchronize(sts[i]));
    std::vector<unsigned long long> tot(4 * (size_t)P);
    HIP_TRY(hipMemcpy(tot.data(), own.d_tot, sizeof(unsigned long long) * tot.size(), hipMemcpyDeviceToHost));
    for (int p = 0; p < P; p++)
        if ((tot[4 * (size_t)p + 2] & 3ull) || (int64_t)tot[4 * (size_t)p] > bound[(size_t)p] || (int64_t)tot[4 * (size_t)p + 1] > bound[(size_t)p]) return abandon();
    for (int p = 0; p < P; p++) {
        PartPlan     &pp = pps[(size_t)p];
        Part         &part = h->parts[(size_t)p];
        const int64_t ns = nsubs[(size_t)p], nzp = d.off[p + 1] - d.off[p];
        pp.tables_on_device = true;
        pp.dev_nchunks = (int64_t)tot[4 * (size_t)p]; pp.dev_nshared = (int64_t)tot[4 * (size_t)p + 1];
        pp.max_nseg = (int64_t)(tot[4 * (size_t)p + 3] >> 1) + (int64_t)(tot[4 * (size_t)p + 3] & 1ull);
        pp.yext = ns + 1 + 2 * pp.dev_nchunks;
        pp.plan.S = pp.S; pp.plan.nz_end = nzp;
        if (pp.phases > 1) {          // no more accumulators than the fullest chunk has rows (+ the dump entry), as plan_part sizes them
            const int64_t most = (int64_t)(tot[4 * (size_t)p + 3] >> 1);
            pp.stage = std::min<int64_t>(pp.stage, std::max<int64_t>(64, (most + 1 + 3) & ~(int64_t)3));
        }
        plan_stage(pp, f32);
        drs[(size_t)p] = DevRows{part.d_rp, 0, nzp, h->stream, &h->plan_ws};
    }
    if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr panels] %d chunk plans as one submission\n", P);
    *done = true;
    return CVR_OK;
}

// what plan_panels_batched left in the parts, released (cvr_create plans the panels again with a longer chunk)
void release_panel_plans(cvr_handle *h)
{
    for (Part &part : h->parts) {
        if (!part.rp_borrowed) for (void *q : {(void *)part.d_rp, (void *)part.d_nzb, (void *)part.d_pad}) if (q) (void)hipFree(q);
        if (!part.tables_borrowed) for (void *q : {(void *)part.img.desc, (void *)part.img.desc2, (void *)part.img.shared}) if (q) (void)hipFree(q);
        part.d_rp = nullptr; part.rp_borrowed = false; part.tables_borrowed = false; part.d_nzb = nullptr; part.d_pad = nullptr; part.img.desc = nullptr; part.img.desc2 = nullptr; part.img.shared = nullptr;
    }
    if (h->panel_rp) { (void)hipFree(h->panel_rp); h->panel_rp = nullptr; }
    if (h->panel_tables) { (void)hipFree(h->panel_tables); h->panel_tables = nullptr; }
}

// device side of one image: allocations and uploads for a planned part (pp = nullptr: plan here, timed into *plan_s)
// (rp == nullptr: part.d_rp is already in place -- the row pointers of a column panel split on the device -- and `dr` describes it)
int build_part(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *ci, const void *va,
                      hipMemcpyKind civa_kind, bool f32, const IOpt &opt, double *plan_s, PartPlan *planned, const DevRows *dr)
{
    const int64_t nz0 = rp ? (nrows ? rp[0] : 0) : dr->nz0, nz1 = rp ? (nrows ? rp[nrows] : 0) : dr->nz1;
    const size_t  vsz = f32 ? 4 : 8, nnz_span = (size_t)nz1;   // arrays are indexed literally from 0
    // the CSR goes to the device first (asynchronously): the automatic layout choice below looks at it there
    const bool adopted = part.d_ci && part.d_va && (rp ? part.d_rp != nullptr : part.csr_borrowed);      // cvr_create's staging copy of the whole CSR, handed over; or a panel's slices of the device split
    if (rp && !adopted) HIP_TRY(hipMalloc(&part.d_rp, sizeof(int64_t) * ((size_t)nrows + 1)));
    if (!adopted) HIP_TRY(hipMalloc(&part.d_ci, sizeof(int32_t) * std::max<size_t>(nnz_span, 1)));
    if (!adopted) HIP_TRY(hipMalloc(&part.d_va, vsz * std::max<size_t>(nnz_span, 1)));
    if (rp && !adopted && nrows > 0) HIP_TRY(hipMemcpyAsync(part.d_rp, rp, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice, h->stream));
    if (nnz_span && !adopted) {
        HIP_TRY(hipMemcpyAsync(part.d_ci, ci, sizeof(int32_t) * nnz_span, civa_kind, h->stream));
        HIP_TRY(hipMemcpyAsync(part.d_va, va, vsz * nnz_span, civa_kind, h->stream));
    }
    PartPlan    local;
    IOpt        popt = opt;
    if (!planned && nrows > 0) {      // a whole matrix (host rows, or a device copy described by `dr`): analysis, plan and conversion as one submission where the resident layout applies
        bool       taken = false;
        const int  rc = build_part_fused(h, part, nrows, ncols, f32, nz0, nz1, opt, popt, &taken);
        if (rc || taken) return rc;
    }
    if (!planned) {
        // (the planner's records come back behind the first kPinnedSmall bytes of the pinned buffer: those belong to the probe and the
        // dictionary scan, which now run at the same time)
        cvr::PlanScratch plan_view = h->plan_ws;
        const bool       shifted = plan_view.pinned && plan_view.pinned_bytes > kPinnedSmall;
        if (shifted) { plan_view.pinned += kPinnedSmall; plan_view.pinned_bytes -= kPinnedSmall; }
        else { plan_view.pinned = nullptr; plan_view.pinned_bytes = 0; }
        struct SyncBack { cvr::PlanScratch &view, &home; ~SyncBack() { home.dev = view.dev; home.dev_bytes = view.dev_bytes; } } sync_back{plan_view, h->plan_ws};      // (the planner may grow its device scratch; also on the error paths)
        DevRows        here{part.d_rp, nz0, nz1, h->stream, &plan_view};
        const bool     on_dev = rp && nrows > 0 && nrows >= device_plan_rows() && !cvr::debug_env("host_plan");      // (the upload of row_ptr is in front of the planner's kernels on the stream)
        const DevRows  via{dr ? dr->rp : nullptr, nz0, nz1, dr ? dr->st : nullptr, &plan_view};      // the caller's device rows, planned through this call's view of the scratch (it may grow: SyncBack)
        const int64_t *prp = on_dev ? nullptr : rp;
        const DevRows *pdr = on_dev ? &here : dr ? &via : nullptr;
        // the plan of the layout the probe is expected to confirm, made while the probe runs (device planner only: it works on the handle's stream)
        PartPlan spec_plan;
        IOpt     spec_opt;
        bool     spec_done = false;
        double   spec_s = 0;
        const std::function<int(const IOpt &)> speculate = [&](const IOpt &so) -> int {
            const double ts = now_s();
            HIP_TRY(plan_part(spec_plan, nrows, ncols, f32, prp, so, pdr));
            spec_opt = so; spec_done = true; spec_s = now_s() - ts;
            return CVR_OK;
        };
        int rc = auto_layout(h, part, nrows, ncols, f32, nz0, nz1, popt, on_dev && !cvr::debug_env("no_speculative_plan") ? &speculate : nullptr);      // (waits for the upload; its own pass is timed into info.probe_s)
        if (rc) return rc;
        rc = choose_hubs(h, part, part.d_ci, nrows, ncols, f32, nz0, nz1, popt, local, true);
        if (rc) return rc;
        const double   t0 = now_s();
        if (plan_s) *plan_s += spec_s;
        if (spec_done && popt.layout_auto_resident && popt.waves_per_block == spec_opt.waves_per_block && popt.steps_per_chunk == spec_opt.steps_per_chunk &&
            popt.x_window == spec_opt.x_window && popt.col_phases == spec_opt.col_phases && local.hub_n == 0)
            local = std::move(spec_plan);
        else
            HIP_TRY(plan_part(local, nrows, ncols, f32, prp, popt, pdr));
        // the resident layout wants every workgroup on a CU of its own at once: one more step per chunk until they fit
        // (longer chunks do not help when it is the cap on a chunk's ROWS -- the LDS accumulators -- that ends them, as with many rows of one
        // or two non-zeros: two steps without fewer chunks and the matrix gets the plain layout instead)
        int64_t before = (int64_t)local.plan.chunks.size();
        int     stuck = 0;
        while (popt.layout_auto_resident && !local.too_large && (int64_t)local.plan.chunks.size() > (int64_t)local.wpb * popt.cus && popt.steps_per_chunk < 4096 && stuck < 2) {
            popt.steps_per_chunk += 4;
            local = PartPlan();
            HIP_TRY(plan_part(local, nrows, ncols, f32, prp, popt, pdr));      // (the resident layout has no hub table: nothing of `local` to keep)
            stuck = (int64_t)local.plan.chunks.size() >= before ? stuck + 1 : 0;
            before = (int64_t)local.plan.chunks.size();
        }
        if (popt.layout_auto_resident && !local.too_large && (int64_t)local.plan.chunks.size() > (int64_t)local.wpb * popt.cus) {
            popt = opt;
            popt.col_phases = 0; popt.layout_auto_resident = 0;
            local = PartPlan();
            HIP_TRY(plan_part(local, nrows, ncols, f32, prp, popt, pdr));
        }
        if (plan_s) *plan_s += now_s() - t0;
        planned = &local;
    }
    PartPlan &pp = *planned;
    if (pp.too_large) return fail(CVR_ERR_INVALID, "matrix too large for 32-bit row ordinals on one GPU");
    const cvr::Plan &plan = pp.plan;
    const int64_t    nchunks = pp.tables_on_device ? pp.dev_nchunks : (int64_t)plan.chunks.size(), yext = pp.yext;
    const int64_t    nshared_total = pp.tables_on_device ? pp.dev_nshared : (int64_t)plan.shared.size();
    const std::vector<uint32_t> &desc = pp.desc, &pad = pp.pad;
    const std::vector<int64_t>  &nzb = pp.nzb;

    part.nrows = nrows; part.nnz = nz1 - nz0; part.nnz_span = nz1; part.nchunks = nchunks; part.nshared = nshared_total; part.yext = yext;
    {
        const int rc = setup_image(h, part, pp, nrows, ncols, f32, nchunks, nshared_total, opt, popt);
        if (rc) return rc;
    }
    cvr::DeviceImage &img = part.img;
    if (!pp.tables_on_device) {          // (else: nzb, pad, desc and the cut rows were written by the planner on the device: plan_panels_batched)
        HIP_TRY(hipMalloc(&part.d_nzb, sizeof(int64_t) * ((size_t)nchunks + 1)));
        HIP_TRY(hipMalloc(&part.d_pad, sizeof(uint32_t) * std::max<size_t>((size_t)nchunks, 1)));
        HIP_TRY(hipMalloc(&img.desc, 16 * std::max<size_t>((size_t)nchunks, 1)));
        HIP_TRY(hipMalloc(&img.shared, 24 * std::max<size_t>(plan.shared.size(), 1)));
    }
    HIP_TRY(hipMalloc(&img.target, 64 * std::max<size_t>((size_t)nchunks, 1)));
    if (pp.phases > 1 && !pp.tables_on_device) {
        HIP_TRY(hipMalloc(&img.desc2, 8 * std::max<size_t>((size_t)nchunks, 1)));
        if (nchunks) HIP_TRY(hipMemcpyAsync(img.desc2, pp.desc2.data(), sizeof(uint32_t) * pp.desc2.size(), hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipMalloc(&img.win_base, sizeof(uint32_t) * ((size_t)nchunks / img.wpb + 1)));
    HIP_TRY(hipMemsetAsync(img.win_base, 0, sizeof(uint32_t) * ((size_t)nchunks / img.wpb + 1), h->stream));
    if (nchunks && !pp.tables_on_device) {
        HIP_TRY(hipMemcpyAsync(part.d_nzb, nzb.data(), sizeof(int64_t) * nzb.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(part.d_pad, pad.data(), sizeof(uint32_t) * pad.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(img.desc, desc.data(), sizeof(uint32_t) * desc.size(), hipMemcpyHostToDevice, h->stream));
    }
    if (!plan.shared.empty() && !pp.tables_on_device)
        HIP_TRY(hipMemcpyAsync(img.shared, plan.shared.data(), sizeof(cvr::Shared) * plan.shared.size(), hipMemcpyHostToDevice, h->stream));
    // narrow chunks (plain layout only): if every chunk spans fewer than 32 767 columns -- banded matrices -- the image stores
    // 16-bit column offsets from the chunk's smallest column: 10 instead of 12 bytes per fp64 slot of a stream-bound SpMV
    // (not for a column panel that will run on one XCD beside seven others: those launches take 32-bit columns, and a panel is wide)
    if (popt.narrow_cols != 0 && nchunks > 0 && img.wpb == 1 && img.win_elems == 0 && img.phases <= 1 && img.hub_n == 0 && !opt.debug_col_mask &&
        !(opt.panel_on_one_xcd && popt.narrow_cols < 0)) {
        uint32_t *d_wide = nullptr, wide = 1;
        HIP_TRY(hipMalloc(&img.cbase, sizeof(uint32_t) * (size_t)nchunks));
        HIP_TRY(hipMalloc(&d_wide, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(d_wide, 0, sizeof(uint32_t), h->stream));
        cvr::DeviceCsr csr;
        csr.col_idx = part.d_ci; csr.nz_begin = part.d_nzb;
        hipError_t e = cvr::launch_chunk_span(img, csr, img.cbase, d_wide, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&wide, d_wide, sizeof(wide), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(d_wide);
        if (e != hipSuccess) return fail(CVR_ERR_HIP, "chunk column spans: %s", hipGetErrorString(e));
        img.c16 = wide == 0;
        if (!img.c16) { (void)hipFree(img.cbase); img.cbase = nullptr; }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));   // the host staging vectors go out of scope; the caller may free its CSR
    return CVR_OK;
}

// second half of build_part, once it is known whether the values go through a dictionary: the stream image
int finish_part(cvr_handle *h, Part &part)
{
    cvr::DeviceImage &img = part.img;
    img.dict = h->d_dict; img.ndict = h->ndict;
    if (img.dict) img.c16 = false;                 // (the dictionary layout keeps 32-bit column words)
    part.stream_bytes = (size_t)part.nchunks * img.G * cvr::group_bytes(img.f32, h->d_dict != nullptr, img.c16, img.tag16);
    // the SpMV kernel's software pipeline issues its stream loads up to 5 groups past the end of a chunk (the buffer
    // descriptor's range check returns zeros for them); the allocation is padded by that much so that the last chunk's
    // run-ahead stays inside it whatever the hardware does with an offset beyond num_records
    const size_t slack = 8 * (size_t)cvr::group_bytes(img.f32, h->d_dict != nullptr, img.c16, img.tag16);
    if (cvr::debug_env("stream_uncached"))   // experiment: matrix image in uncached (MTYPE UC) memory, so that it cannot displace x in L2
        HIP_TRY(hipExtMallocWithFlags((void **)&img.stream, part.stream_bytes + slack, hipDeviceMallocUncached));
    else
        HIP_TRY(hipMalloc(&img.stream, part.stream_bytes + slack));
    // gang chunks: only the groups that hold a gang's non-zeros are written by the converter; the rest of its chunks' allocations is never read, but an
    // image is a deterministic function of its matrix (the image cache, the mirror's bits): zeros
    if (img.gang) HIP_TRY(hipMemsetAsync(img.stream, 0, part.stream_bytes + slack, h->stream));
    if (img.gang && !img.tag16) {          // gang chunks: the first column of every group (zeros behind a gang's last group and behind the last gang: the kernel's ring runs ahead)
        const size_t nb = sizeof(uint32_t) * ((size_t)part.nchunks * img.G + 4096);
        HIP_TRY(hipMalloc(&img.gbase, nb));
        HIP_TRY(hipMemsetAsync(img.gbase, 0, nb, h->stream));
    }
    return CVR_OK;
}


}  // namespace cvrh

// cvr_fused.hip -- preprocessing as ONE submission, for the matrices the automatic layout makes "resident" (web-Google-sized: every
// chunk on the chip at once, LDS window of x, column phases).  The reference's pre_processing (spmv.cpp:565-1014) is one host pass;
// the staged path here (cvr_layout.hip, cvr_preprocess) is a chain of small device passes with the host in between -- probe ->
// decide -> plan -> records to the host -> tables to the device -> segment table -> conversion -- and on a matrix of this size the
// synchronisations, the copies and the launches between them cost more than the kernels (profiles/r03_pre_timeline.txt).  Here:
//
//   handle's stream   planner kernels (the last writes desc / desc2 / pad / nzb on the device) -> segment table -> conversion
//   side stream 1     layout probe                          } the host waits for these two only, while the planner runs, confirms
//   side stream 0     dictionary scan, later the windows    } the layout the plan was made for and picks the converter's variant
//
// and one synchronisation at the end.  Every kernel behind the planner reads the number of chunks on the device; the buffers have
// room for as many chunks as the resident layout allows at all (workgroups <= CUs): a plan with more falls back to the staged path,
// which lengthens the chunks, as does a probe that does not confirm the layout.  The image is the staged path's, bit for bit
// (tests/test_gpu_parity.py::test_fused_preprocessing_same_image).
#include "cvr_internal.h"

using namespace cvrh;

namespace cvrh {

namespace {

// everything the attempt allocates: handed to the part on success, released otherwise
struct Attempt {
    cvr::DeviceImage img{};
    int64_t  *d_nzb = nullptr;
    uint32_t *d_pad = nullptr;
    void     *arena = nullptr;
    void     *d_dict = nullptr;
    uint8_t  *d_codes = nullptr;        // the values as dictionary codes (conversion only)
    bool      keep = false;
    ~Attempt()
    {
        if (keep) return;
        for (void *p : {(void *)img.stream, (void *)img.desc, (void *)img.desc2, (void *)img.target, (void *)img.win_base, (void *)d_nzb, (void *)d_pad, arena, d_dict, (void *)d_codes})
            if (p) (void)hipFree(p);
    }
};

}  // namespace

// CVR_FUSED_TRACE=1: why the one-submission path was not taken, on stderr
#define NOT_TAKEN(why) do { if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr fused] not taken: %s\n", why); return CVR_OK; } while (0)

int build_part_fused(cvr_handle *h, Part &part, int64_t nrows, int64_t ncols, bool f32, int64_t nz0, int64_t nz1, const IOpt &opt, IOpt &popt, bool *taken)
{
    *taken = false;
    if (opt.interleave > 0) NOT_TAKEN("interleaved chunks");
    const int64_t nnz = nz1 - nz0, vs = f32 ? 4 : 8;
    // the matrices of auto_layout's resident form, planned on the device, with the probe's usual answer assumed
    if (opt.steps_per_chunk != 0 || opt.waves_per_block != 0 || opt.x_window >= 0 || opt.col_phases >= 0 || opt.debug_col_mask) NOT_TAKEN("layout options given");
    if (cvr::debug_env("no_auto_layout") || cvr::debug_env("no_fused") || cvr::debug_env("no_speculative_plan") || cvr::debug_env("host_plan") || cvr::debug_env("seg_by_rows")) NOT_TAKEN("switched off by the environment");
    if (nrows < 4096 || ncols < 4096 || nrows < device_plan_rows() || nnz <= 0 || nz1 >= (int64_t)0x7fffffff00ll) NOT_TAKEN("too small for the device planner");
    const double xbytes = (double)ncols * vs;
    if (xbytes <= 2.5e6) NOT_TAKEN("x fits the L2s: no column phases");
    const double slots = ((double)nnz + (double)nrows / 4) * 1.006;
    int          best_w = 0, best_S = 0;
    if (!resident_candidate(slots, opt.cus, &best_w, &best_S)) NOT_TAKEN("no resident layout for this size");
    if (!h->plan_ws.pinned || h->plan_ws.pinned_bytes < kPinnedSmall + (64 << 10)) NOT_TAKEN("no pinned buffer");
    hipStream_t pstream = side_stream(h->device, 1), side = side_stream(h->device, 0);
    if (!pstream || !side) NOT_TAKEN("no side streams");

    IOpt spec = opt;
    spec.layout_auto_resident = 1; spec.waves_per_block = best_w; spec.steps_per_chunk = best_S;
    spec.x_window = (int32_t)((96 * 1024) / vs);
    spec.col_phases = (int32_t)std::min(32.0, std::max(2.0, std::floor(xbytes / 450e3 + 0.5)));
    PartPlan pp;
    pp.S = best_S;
    const int64_t max_rows = plan_layout(pp, ncols, f32, spec);
    if (pp.phases <= 1 || pp.lds_short || pp.win <= 0) NOT_TAKEN("LDS budget leaves no phases or window");
    const int64_t cap = (int64_t)cvr::kLanes * pp.S, room = (int64_t)pp.wpb * opt.cus;        // more chunks than this: the staged path lengthens them
    if (nrows + 1 + 2 * room >= (int64_t)0xffffffffu) NOT_TAKEN("row ordinals");

    Attempt at;
    cvr::DeviceImage &img = at.img;
    {
        Part scratch;           // (setup_image fills a Part's image; a Part does not release anything by itself)
        const int rc = setup_image(h, scratch, pp, nrows, ncols, f32, room, 0, opt, spec);
        img = scratch.img;
        if (rc) return rc;
    }
    if (!cvr::seg_table_packed_ok(img)) NOT_TAKEN("chunks too long for the packed segment table");
    const bool   with_dict = opt.value_dict != 0;
    const size_t gb_plain = (size_t)cvr::group_bytes(f32, false, false, img.tag16);
    auto         up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t n1 = (size_t)room * (size_t)cap;
    const size_t o_begin = 0, o_cnt = o_begin + up(sizeof(int64_t) * n1), o_flags = o_cnt + up(sizeof(uint32_t) * ((size_t)room + 1)), arena_bytes = o_flags + 256;
    // (buffers for as many chunks as the layout allows; if the device has no room for them the staged path, which allocates what the plan needs, takes over)
    bool ok = true;
    auto alloc = [&ok](auto **p, size_t bytes) { if (ok && hipMalloc(reinterpret_cast<void **>(p), bytes) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; ok = false; } };
    alloc(&at.d_nzb, sizeof(int64_t) * ((size_t)room + 1));
    alloc(&at.d_pad, sizeof(uint32_t) * (size_t)room);
    alloc(&img.desc, 16 * (size_t)room);
    alloc(&img.desc2, 8 * (size_t)room);
    alloc(&img.target, 64 * (size_t)room);
    alloc(&img.win_base, sizeof(uint32_t) * ((size_t)room / img.wpb + 1));
    alloc(&img.stream, (size_t)room * img.G * gb_plain + 8 * gb_plain);      // (room for either form of the values; the staged path's slack behind the last chunk)
    alloc(&at.arena, arena_bytes);
    if (with_dict) alloc(&at.d_dict, (size_t)vs * cvr::kDictMax);
    if (with_dict && !cvr::debug_env("no_dict_codes")) alloc(&at.d_codes, (size_t)nz1);
    if (!ok) NOT_TAKEN("no device memory for the attempt's buffers");
    cvr::SegTable seg;
    uint8_t      *res_dev = nullptr;
    {
        uint8_t *a = static_cast<uint8_t *>(at.arena);
        seg.begin = reinterpret_cast<int64_t *>(a + o_begin); seg.cnt = reinterpret_cast<uint32_t *>(a + o_cnt); 
        res_dev = a + o_flags;                 // 64 bytes: totals [4 x u64] | segment-table flags [2 x u32] | segments | converter flags
        seg.flags = reinterpret_cast<uint32_t *>(res_dev + 32);
    }
    HIP_TRY(hipMemsetAsync(img.win_base, 0, sizeof(uint32_t) * ((size_t)room / img.wpb + 1), h->stream));
    HIP_TRY(hipMemsetAsync(res_dev, 0, 64, h->stream));
    if (at.d_dict) HIP_TRY(hipMemsetAsync(at.d_dict, 0, (size_t)vs * cvr::kDictMax, h->stream));
    if (h->events.size() < 2) NOT_TAKEN("no events");
    // events: the plan is on the device (the windows wait for it on their stream); the probe's / the dictionary scan's results are on the
    // host; the segment table is written (its total is summed beside the conversion); the side stream's part of the chain is done
    struct Events {
        hipEvent_t planned = nullptr, probed = nullptr, scanned = nullptr, seg_done = nullptr, side_done = nullptr, coded = nullptr;
        ~Events() { for (hipEvent_t e : {planned, probed, scanned, seg_done, side_done, coded}) if (e) (void)hipEventDestroy(e); }
    } ev;
    for (hipEvent_t *e : {&ev.planned, &ev.probed, &ev.scanned, &ev.seg_done, &ev.side_done, &ev.coded}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    const hipEvent_t planned = ev.planned;

    // pinned: probe output | dictionary table | flags (the first kPinnedSmall bytes, as in auto_layout) | results of the chain | dictionary values
    uint8_t            *pin = h->plan_ws.pinned;
    unsigned long long *outv = reinterpret_cast<unsigned long long *>(pin + kPinnedProbe), *tabv = reinterpret_cast<unsigned long long *>(pin + kPinnedDictTab);
    uint32_t           *flagv = reinterpret_cast<uint32_t *>(pin + kPinnedDictFlags);
    unsigned long long *res_totals = reinterpret_cast<unsigned long long *>(pin + kPinnedSmall);      // the result block: [4] totals,
    uint32_t           *res_small = reinterpret_cast<uint32_t *>(pin + kPinnedSmall + 32);             // segment-table flags [2], segments, converter flags
    uint8_t            *dict_host = pin + kPinnedSmall + 256;                                          // [kDictMax] values
    unsigned long long *d_probe = reinterpret_cast<unsigned long long *>(h->d_small + kSmallProbe);

    HIP_TRY(hipStreamSynchronize(h->stream));          // the upload (and the memsets above)
    const double t0 = now_s();
    // ---- submission: the critical path first (planner -> segment table on the handle's stream), then the two analysis passes the host
    // waits for (probe: side stream 1; dictionary scan: side stream 0), then the side work behind them (windows, sum of the segment counts).
    // Measured orders (tools/wg_create_once.py, CVR_FUSED_TRACE): this one ends after 189-191 us; the dictionary scan first and the
    // codes pass as early as its result allows 199-215 (everything then runs beside the segment table, which takes twice as long for it);
    // the analysis passes in front of the planner 231-240 (the host needs ~4 us per call: the planner starts 25 us late).
    hipError_t e = hipEventRecord(h->events[0], h->stream);
    cvr::DevicePlan dp;
    cvr::PlanTables tables;
    tables.desc = img.desc; tables.desc2 = img.desc2; tables.pad = at.d_pad; tables.nzb = at.d_nzb; tables.room = (uint32_t)room; tables.phased = true;
    tables.totals = reinterpret_cast<unsigned long long *>(res_dev);
    if (e == hipSuccess) e = cvr::plan_chunks_device_enqueue(part.d_rp, nrows, nz1, pp.S, opt.split_threshold, max_rows, h->stream, &h->plan_ws, &dp, &tables);
    if (e == hipSuccess && !dp.declined) e = hipEventRecord(planned, h->stream);
    // the segment table needs nothing the host decides below (a layout the probe does not confirm wastes it): straight behind the planner
    const uint32_t *nch_dev = reinterpret_cast<const uint32_t *>(dp.totals);      // (little endian: the low half of totals[0])
    cvr::DeviceCsr  csr;
    csr.row_ptr = part.d_rp; csr.col_idx = part.d_ci; csr.vals = part.d_va; csr.nz_begin = at.d_nzb; csr.pad_cnt = at.d_pad;
    const bool go = e == hipSuccess && !dp.declined;
    if (go) e = cvr::launch_seg_build(img, csr, seg, h->stream, nch_dev, false);
    if (go && e == hipSuccess && !seg.packed) e = hipErrorUnknown;      // (seg_table_packed_ok above)
    if (go && e == hipSuccess) e = hipEventRecord(ev.seg_done, h->stream);
    if (e == hipSuccess) e = cvr::launch_probe(part.d_rp, part.d_ci, nrows, ncols, (uint32_t)(spec.x_window / 4), d_probe, pstream, h->small_clean);
    if (e == hipSuccess) e = hipMemcpyAsync(outv, d_probe, sizeof(unsigned long long) * 2 * cvr::kProbeBlocks, hipMemcpyDeviceToHost, pstream);
    if (e == hipSuccess) e = hipEventRecord(ev.probed, pstream);
    if (e == hipSuccess && with_dict) e = enqueue_dict_scan(h, part.d_va, nz0, nz1, f32, true, tabv, flagv, true, side);
    if (e == hipSuccess) e = hipEventRecord(ev.scanned, side);
    h->small_clean = false;
    if (go && e == hipSuccess) e = hipStreamWaitEvent(side, planned, 0);
    if (go && e == hipSuccess) e = cvr::launch_window(img, csr, side, nch_dev);
    if (go && e == hipSuccess) e = hipStreamWaitEvent(side, ev.seg_done, 0);
    if (go && e == hipSuccess) e = cvr::launch_seg_total(seg, (uint32_t)room, nch_dev, reinterpret_cast<uint32_t *>(res_dev + 40), side);
    if (go && e == hipSuccess) e = hipEventRecord(ev.side_done, side);
    const double t_sub1 = now_s();
    // ---- the host looks at the probe and the dictionary while planner and segment table run
    if (e == hipSuccess) e = hipEventSynchronize(ev.probed);
    if (e == hipSuccess) e = hipEventSynchronize(ev.scanned);
    if (e != hipSuccess || dp.declined) {
        (void)hipStreamSynchronize(pstream); (void)hipStreamSynchronize(side); (void)hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return fail(CVR_ERR_HIP, "fused preprocessing (analysis): %s", hipGetErrorString(e));
        NOT_TAKEN("device planner declined");
    }
    unsigned long long out[2] = {0, 0};
    for (uint32_t b = 0; b < cvr::kProbeBlocks; b++) { out[0] |= outv[2 * b]; out[1] += outv[2 * b + 1]; }
    const bool   sorted = out[0] == 0;
    const double near = (double)out[1] / std::max<double>((double)nnz, 1.0);
    h->info.near_diagonal_share = near;
    const bool confirmed = sorted && near >= 0.15 && near < 0.9;       // window and phases, as auto_layout decides them
    const double t_probe = now_s();
    uint32_t ndict = 0;
    if (confirmed && with_dict && !(flagv[0] & 1u)) {
        std::vector<unsigned long long> d;
        d.push_back(0);                                                  // +0.0: the value of every pad slot
        for (uint32_t i = 0; i < 1024; i++) if (tabv[i] != ~0ull) d.push_back(tabv[i]);
        if (flagv[0] & 2u) d.push_back(f32 ? 0xffffffffull : ~0ull);     // the all-ones pattern occurs as a value
        std::sort(d.begin(), d.end());
        d.erase(std::unique(d.begin(), d.end()), d.end());
        if (d.size() <= (size_t)cvr::kDictMax) {
            ndict = (uint32_t)d.size();
            if (f32) { uint32_t *q = reinterpret_cast<uint32_t *>(dict_host); for (uint32_t i = 0; i < ndict; i++) q[i] = (uint32_t)d[i]; }
            else memcpy(dict_host, d.data(), sizeof(unsigned long long) * ndict);
        }
    }
    const double t_dict = now_s();
    if (!confirmed) {
        (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(side); (void)hipStreamSynchronize(pstream);
        NOT_TAKEN("probe did not confirm window + phases");
    }
    // ---- the rest of the chain: the conversion, in the variant the dictionary scan decides.  With a dictionary the values are first
    // rewritten as codes (one coalesced pass on the probe's stream, idle by now, beside the end of the segment table): the converter then
    // reads a byte per value and searches nothing.  It reads the dictionary itself from the pinned host buffer (a few values per
    // workgroup); the copy the SpMV kernel will use goes to the device in front of the codes pass.
    if (ndict) e = hipMemcpyAsync(at.d_dict, dict_host, (size_t)vs * ndict, hipMemcpyHostToDevice, pstream);
    img.dict = ndict ? dict_host : nullptr; img.ndict = ndict;
    if (ndict && at.d_codes) {
        if (e == hipSuccess) e = cvr::launch_dict_codes(part.d_va, nz0, nz1, f32, at.d_dict, ndict, at.d_codes, reinterpret_cast<uint32_t *>(res_dev + 44), pstream);
        if (e == hipSuccess) e = hipEventRecord(ev.coded, pstream);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, ev.coded, 0);
        csr.codes = at.d_codes;
    }
    if (e == hipSuccess) e = cvr::launch_convert(img, csr, reinterpret_cast<uint32_t *>(res_dev + 44), h->stream, &seg, nch_dev);
    img.dict = ndict ? at.d_dict : nullptr;
    if (e == hipSuccess) e = hipEventRecord(h->events[1], h->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, ev.side_done, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(res_totals, res_dev, 64, hipMemcpyDeviceToHost, h->stream);
    const double t_sub2 = now_s();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(pstream);
    const double t_end = now_s();
    if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr fused] host: first submission %.0f us, probe + scan waited for until %.0f, dictionary %.0f, second submission %.0f, end %.0f\n", (t_sub1 - t0) * 1e6, (t_probe - t0) * 1e6, (t_dict - t0) * 1e6, (t_sub2 - t0) * 1e6, (t_end - t0) * 1e6);
    if (e != hipSuccess) { (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(side); (void)hipStreamSynchronize(pstream); return fail(CVR_ERR_HIP, "fused preprocessing: %s", hipGetErrorString(e)); }
    const int64_t nchunks = (int64_t)res_totals[0], nshared = (int64_t)res_totals[1], most = (int64_t)(res_totals[3] >> 1);
    if ((res_totals[2] & 3ull) || nchunks > room || nshared > dp.bound) NOT_TAKEN("more chunks than workgroup slots (or a row block beyond 32-bit slots)");      // the staged path lengthens the chunks
    if (res_small[0] & 2u) NOT_TAKEN("a chunk beyond the launch's length");                                                     // (a chunk the launch has no room for: cannot happen, the plan keeps to S)
    if (res_small[0] & 1u) return fail(CVR_ERR_INVALID, "col_phases needs the column indices of every row in ascending order");
    if (res_small[3]) return fail(CVR_ERR_INTERNAL, "device converter self-check failed (flags 0x%x)", res_small[3]);

    // ---- the part takes the image over
    img.nchunks = (uint32_t)nchunks;
    img.nshared = (uint32_t)nshared;
    img.ystage = (uint32_t)std::min<int64_t>(pp.stage, std::max<int64_t>(64, (most + 1 + 3) & ~(int64_t)3));      // no more accumulators than the fullest chunk has rows (+ the dump entry)
    HIP_TRY(hipMalloc(&img.shared, 24 * std::max<size_t>((size_t)nshared, 1)));
    if (nshared) HIP_TRY(hipMemcpyAsync(img.shared, dp.shared, sizeof(cvr::Shared) * (size_t)nshared, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));       // (the planner's scratch is released before cvr_create returns)
    at.keep = true;
    part.img = img;
    part.d_nzb = at.d_nzb; part.d_pad = at.d_pad;
    part.nrows = nrows; part.nnz = nnz; part.nnz_span = nz1; part.nchunks = nchunks; part.nshared = nshared; part.yext = nrows + 1 + 2 * nchunks;
    part.stream_bytes = (size_t)nchunks * img.G * cvr::group_bytes(f32, ndict != 0, false, img.tag16);
    h->d_dict = ndict ? at.d_dict : nullptr;
    if (!ndict && at.d_dict) (void)hipFree(at.d_dict);
    h->ndict = ndict;
    h->dict_scanned = true;
    (void)hipFree(at.arena);
    if (at.d_codes) (void)hipFree(at.d_codes);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->events[0], h->events[1]));
    cvr_info &in = h->info;
    in.convert_s = ms * 1e-3;
    in.nsegments = res_small[2];
    in.probe_s = t_probe - t0;
    in.dict_s = t_dict - t_probe;
    in.plan_s += t_end - t_dict;            // planner, segment table and conversion: one stretch of the stream (convert_s = the last two on the device)
    in.preprocess_wall_s = 0;
    in.preprocess_fused = 1;
    h->converted = true;
    h->preconverted = true;
    popt = spec;
    *taken = true;
    return CVR_OK;
}

}  // namespace cvrh

// cvr_capi.hip -- the C ABI of include/cvr_amd.h: handle life cycle, device memory, timing.
// Host orchestration that the reference keeps in main() (allocation block spmv.cpp:1777-1829, calls at
// spmv.cpp:1857 and 1882) lives behind the handle here; the caller keeps only CSR, x and y.
#include "cvr_internal.h"

using namespace cvrh;

namespace cvrh {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}



// y_ext = A x for the whole handle on `st`: one SpMV launch, or one per column panel followed by the combine
hipError_t run_spmv(cvr_handle *h, const void *x, void *y, hipStream_t st)
{
    // A buffer of the handle that every SpMV writes and reads (the panels' partial sums h->d_z; the hub table's compacted copy of x): a launch on ANOTHER stream
    // must not start before the previous one is through with it.  The event is recorded when the stream CHANGES, on the stream of the launches before (behind
    // their last pass), not after every SpMV: back-to-back launches on one stream are ordered by the stream, and an event per SpMV is a packet between the
    // combine pass and the next panel kernel (profiles/r06_z_event.log).
    static const bool event_per_spmv = cvr::debug_env("z_event_per_spmv");
    auto enter = [&]() -> hipError_t {
        if (!h->z_used) return hipSuccess;
        if (event_per_spmv) return hipStreamWaitEvent(st, h->z_free, 0);
        if (h->z_stream == st) return hipSuccess;
        // (stream capture: a dependency between a capturing stream and one outside the capture cannot be expressed -- and recording on the legacy stream would
        //  end the capture; the graph's replays are ordered by the stream they are launched into.  Two streams of ONE capture do get their edge.)
        hipStreamCaptureStatus cap_now = hipStreamCaptureStatusNone, cap_before = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap_now);
        if (hipStreamIsCapturing(h->z_stream, &cap_before) != hipSuccess) { (void)hipGetLastError(); cap_before = hipStreamCaptureStatusNone; }
        if ((cap_now == hipStreamCaptureStatusActive) != (cap_before == hipStreamCaptureStatusActive)) return hipSuccess;
        if (hipEventRecord(h->z_free, h->z_stream) != hipSuccess) {      // (that stream was destroyed meanwhile: whatever it still runs is waited for the blunt way)
            (void)hipGetLastError();
            return hipDeviceSynchronize();
        }
        return hipStreamWaitEvent(st, h->z_free, 0);
    };
    auto leave = [&]() -> hipError_t {
        h->z_used = true;
        h->z_stream = st;
        return event_per_spmv ? hipEventRecord(h->z_free, st) : hipSuccess;
    };
    if (!h->paneled()) {
        if (h->parts.empty()) return hipSuccess;
        if (h->parts[0].img.hub_n == 0) return cvr::launch_spmv(h->parts[0].img, x, y, st);
        hipError_t e = enter();
        if (e != hipSuccess) return e;
        e = cvr::launch_spmv(h->parts[0].img, x, y, st);
        if (e != hipSuccess) return e;
        return leave();
    }
    { const hipError_t e = enter(); if (e != hipSuccess) return e; }
    if (h->d_multi) {          // eight panels per launch, panel b & 7 on the XCD of the workgroups b
        cvr::DeviceImage shared = h->parts[0].img;
        shared.ystage = h->multi_ystage;
        shared.flip_now = shared.ilv_flip ? (h->spmv_calls++ & 1u) : 0u;
        uint32_t most = 0;
        for (uint32_t c : h->multi_chunks) most = std::max(most, c);
        if (most && h->d_fuse) {      // gang chunks: the workgroup that completes a block of rows adds its partial sums and writes y -- no combine launch
            hipError_t e = cvr::launch_spmv(shared, x, nullptr, st, false, h->d_multi, most, (uint32_t)h->multi_chunks.size(), nullptr, h->d_fuse, y);
            if (e != hipSuccess) return e;
            if (h->max_nshared) {          // rows cut over chunks: their carries summed as before, then those rows' y once more
                e = cvr::launch_fixup_multi(h->d_fixparts, (uint32_t)h->parts.size(), h->max_nshared, h->vsz == 4, st);
                if (e == hipSuccess) e = cvr::launch_fuse_patch(h->d_fuse_cut, h->fuse_ncut, h->d_fuse_panels, h->d_cpanels, h->d_fuse_nsub, (uint32_t)h->parts.size(), y, h->vsz == 4, st);
                if (e != hipSuccess) return e;
            }
            return leave();
        }
        if (most) {      // all rounds in one grid
            const hipError_t e = cvr::launch_spmv(shared, x, nullptr, st, false, h->d_multi, most, (uint32_t)h->multi_chunks.size());
            if (e != hipSuccess) return e;
        }
    } else
        for (const Part &p : h->parts) {
            hipError_t e = cvr::launch_spmv(p.img, x, static_cast<uint8_t *>(h->d_z) + (size_t)p.zoff * h->vsz, st, false);
            if (e != hipSuccess) return e;
        }
    // rows cut over chunks: their carries are summed by a launch in front of the combine pass -- or, a handful of them and the pass in its bitmap form, inside it
    const bool fold = h->d_cbits && h->d_cut && h->ncut_fold && h->combine_mul == 1 && h->parts.size() <= 16;
    hipError_t e = fold ? hipSuccess : cvr::launch_fixup_multi(h->d_fixparts, (uint32_t)h->parts.size(), h->max_nshared, h->vsz == 4, st);
    if (e != hipSuccess) return e;
    e = cvr::launch_combine(h->d_cpanels, (uint32_t)h->parts.size(), h->d_block_off, y, (uint32_t)h->info.nrows, h->vsz == 4, st, h->combine_batch, h->combine_mul, h->d_cbits, fold ? h->d_cut : nullptr,
                            fold ? h->ncut_fold : 0u);
    if (e != hipSuccess) return e;
    return leave();
}

int setup_combine_bits(cvr_handle *h, int64_t nsub)
{
    if (!h->paneled() || !h->d_cpanels || !h->d_block_off || h->info.nrows <= 0 || h->d_cbits) return CVR_OK;
    const size_t   P = h->parts.size();
    const uint32_t nrows = (uint32_t)h->info.nrows, nblocks = (nrows + cvr::kCombineRows - 1) / cvr::kCombineRows;
    // The bitmap form issues a load instruction per (row, panel) whether the sum exists or not: it pays where most of them do -- half or more of the (row, panel)
    // pairs filled.  com-Orkut shape (6.5 sums per row in 8 panels: 81 %) 574 -> 566 us; soc-LiveJournal1 shape (2.4 in 16: 15 %) 206 -> 231, wiki-Talk (0.11 in 8)
    // 32.7 -> 39.4: those keep their row numbers (profiles/r06_combine_bitmap.log)
    bool on = P <= 16 && nsub * 2 >= (int64_t)P * (int64_t)nrows;
    if (const char *e = cvr::debug_env("combine_bits")) on = atoi(e) != 0 && P <= 16;
    if (!on) return CVR_OK;
    if (hipMalloc(&h->d_cbits, sizeof(uint32_t) * 32 * (size_t)nblocks * P) != hipSuccess) { (void)hipGetLastError(); h->d_cbits = nullptr; return CVR_OK; }      // (no memory for it: the row numbers do)
    if (cvr::launch_combine_bits_build(h->d_cpanels, (uint32_t)P, h->d_block_off, nrows, h->d_cbits, h->stream) != hipSuccess) return CVR_ERR_HIP;
    h->info.image_bytes += (int64_t)(sizeof(uint32_t) * 32 * (size_t)nblocks * P);
    // the rows cut over chunks, when they are few: folded into the pass (CutEntry, cvr_kernels.h) -- the com-Orkut shape's two rows cost a launch of 4.7 us per SpMV
    int64_t cut = 0;
    for (const Part &p : h->parts) cut += p.nshared;
    if (cut > 0 && cut <= (int64_t)cvr::kMaxCutFold && h->d_fixparts && h->d_rows && h->d_rows16 && !cvr::debug_env("no_cut_fold")) {
        void *m = nullptr;
        if (hipMalloc(&m, sizeof(cvr::CutEntry) * cvr::kMaxCutFold + 16) != hipSuccess) { (void)hipGetLastError(); return CVR_OK; }
        h->d_cut = static_cast<cvr::CutEntry *>(m);
        uint32_t *count = reinterpret_cast<uint32_t *>(h->d_cut + cvr::kMaxCutFold);
        if (hipMemsetAsync(m, 0, sizeof(cvr::CutEntry) * cvr::kMaxCutFold + 16, h->stream) != hipSuccess) return CVR_ERR_HIP;
        if (cvr::launch_cut_table(h->d_fixparts, (uint32_t)P, h->max_nshared, h->d_cpanels, h->d_rows16, h->d_rows, h->d_cut, count, h->stream) != hipSuccess) return CVR_ERR_HIP;
        h->ncut_fold = (uint32_t)cut;
    }
    if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] combine pass: bitmap form (%zu panels, %lld partial sums for %u rows), %u cut rows folded into it\n", P, (long long)nsub, nrows, h->ncut_fold);
    return CVR_OK;
}

// The fused combine (cvr_kernels.h: FuseArgs) for a handle whose panels all carry gang chunks and run one per XCD: the gangs' block ranges, the blocks'
// expected counts, the counters, the panels' table for the set-up kernels and the list of rows cut over chunks -- one allocation, made on the handle's
// stream behind whatever wrote the panels' chunk tables.  CVR_DEBUG=no_fuse keeps the combine pass a launch of its own.
int setup_fuse(cvr_handle *h)
{
    // MEASURED AND NOT ADOPTED (round 6; CVR_DEBUG=fuse switches it on, the parity tests run it): a block of rows is complete only when the LAST of its
    // panels has delivered, i.e. during the second round of panels, and all ~130 blocks of a gang's row range complete at the same workgroup -- a few dozen
    // workgroups chip-wide then add 178 MB of partial sums at one CU's rate each while the others wait for nothing: soc-LiveJournal1 shape 1 796 us against
    // 210 with the pass as a launch of its own, com-Orkut shape 1 208 against 588 (profiles/r06_fused_combine_not_adopted.log).
    if (!cvr::debug_env("fuse")) return CVR_OK;
    if (!h->paneled() || !h->d_multi || !h->d_cpanels || !h->d_block_off || !h->d_rows || h->info.nrows <= 0) return CVR_OK;
    const uint32_t P = (uint32_t)h->parts.size(), gw = h->parts[0].img.gang;
    if (!gw || P > (uint32_t)cvr::kMaxSplitPanels) return CVR_OK;
    uint32_t ngangs = 0, ncut = 0;
    std::vector<cvr::FusePanel> fp(P);
    std::vector<uint32_t>       nsub(P), cut0(P);
    int64_t roff = 0;
    for (uint32_t p = 0; p < P; p++) {
        const Part &q = h->parts[p];
        if (q.img.gang != gw || !q.img.desc2) return CVR_OK;
        fp[p] = cvr::FusePanel{q.img.desc, q.img.desc2, h->d_rows + roff, q.img.nchunks, ngangs};
        nsub[p] = (uint32_t)q.nrows; cut0[p] = ncut;
        roff += q.nrows;
        ngangs += (q.img.nchunks + gw - 1) / gw;
        ncut += (uint32_t)q.nshared;
    }
    if (!ngangs) return CVR_OK;
    const uint32_t nblocks = (uint32_t)((h->info.nrows + cvr::kCombineRows - 1) / cvr::kCombineRows);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_args = 0, o_cnt = o_args + up(sizeof(cvr::FuseArgs)), o_exp = o_cnt + up(4 * (size_t)nblocks), o_rng = o_exp + up(4 * (size_t)nblocks), o_fl = o_rng + up(8 * (size_t)ngangs),
                 o_pan = o_fl + up(8 * ((size_t)ngangs + 1)), o_nsub = o_pan + up(sizeof(cvr::FusePanel) * P), o_cut = o_nsub + up(4 * (size_t)P), total = o_cut + up(4 * (size_t)std::max(ncut, 1u));
    HIP_TRY(hipMalloc(&h->fuse_mem, total));
    uint8_t *m = static_cast<uint8_t *>(h->fuse_mem);
    cvr::FuseArgs fa;
    fa.cnt = reinterpret_cast<uint32_t *>(m + o_cnt); fa.expect = reinterpret_cast<uint32_t *>(m + o_exp); fa.range = reinterpret_cast<uint2 *>(m + o_rng);
    fa.panels = h->d_cpanels; fa.block_off = h->d_block_off; fa.npanels = P; fa.nblocks = nblocks; fa.nrows = (uint32_t)h->info.nrows; fa.ngangs = ngangs;
    h->d_fuse_panels = reinterpret_cast<cvr::FusePanel *>(m + o_pan);
    h->d_fuse_nsub = reinterpret_cast<uint32_t *>(m + o_nsub);
    h->d_fuse_cut = reinterpret_cast<uint32_t *>(m + o_cut);
    h->fuse_ncut = ncut;
    HIP_TRY(hipMemcpyAsync(m + o_args, &fa, sizeof(fa), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_fuse_panels, fp.data(), sizeof(cvr::FusePanel) * P, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_fuse_nsub, nsub.data(), 4 * (size_t)P, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(m + o_fl, 0, 8 * ((size_t)ngangs + 1), h->stream));
    HIP_TRY(cvr::launch_fuse_setup(h->d_fuse_panels, P, gw, ngangs, nblocks, reinterpret_cast<uint2 *>(m + o_rng), reinterpret_cast<uint32_t *>(m + o_fl), reinterpret_cast<uint32_t *>(m + o_exp), fa.cnt, h->stream));
    for (uint32_t p = 0; p < P; p++) HIP_TRY(cvr::launch_fuse_cut_rows(h->parts[p].img.shared, (uint32_t)h->parts[p].nshared, fp[p].rows, h->d_fuse_cut + cut0[p], h->stream));
    // the panels' launch table: every panel's first gang in this numbering
    std::vector<cvr::PanelArgs> pa(h->multi_chunks.size() * 8);
    HIP_TRY(hipStreamSynchronize(h->stream));          // (the staging vectors above go out of scope)
    HIP_TRY(hipMemcpy(pa.data(), h->d_multi, sizeof(cvr::PanelArgs) * pa.size(), hipMemcpyDeviceToHost));
    for (uint32_t p = 0; p < P; p++) { const int32_t s = h->parts[p].multi_slot; if (s < 0 || (size_t)s >= pa.size()) { (void)hipFree(h->fuse_mem); h->fuse_mem = nullptr; return CVR_OK; } pa[(size_t)s].gang0 = fp[p].gang0; }
    HIP_TRY(hipMemcpy(h->d_multi, pa.data(), sizeof(cvr::PanelArgs) * pa.size(), hipMemcpyHostToDevice));
    h->d_fuse = reinterpret_cast<cvr::FuseArgs *>(m + o_args);
    return CVR_OK;
}

IOpt make_iopt(const cvr_options *in)
{
    IOpt o;
    if (in) static_cast<cvr_options &>(o) = *in; else cvr_default_options(&o);
    auto env = [](const char *name) { const char *e = getenv(name); return e ? (int32_t)strtol(e, nullptr, 0) : 0; };
    o.debug_col_mask = env("CVR_DEBUG_COL_MASK");
    return o;
}

Chip chip_of(int device)
{
    static std::mutex mu;
    static Chip       cache[64];
    static bool       known[64] = {};
    if (device < 0 || device >= 64) return Chip{};
    std::lock_guard<std::mutex> lk(mu);
    if (!known[device]) {
        hipDeviceProp_t prop;
        Chip            c;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            c.cus = prop.multiProcessorCount;
            c.xcds = std::max(1, c.cus / 32);
        } else (void)hipGetLastError();
        cache[device] = c; known[device] = true;
    }
    return cache[device];
}


// Interleaved images whose stream does not stay in the Infinity Cache from one SpMV to the next (256 MiB: the image, x and the partial sums share it)
// get helper wavefronts -- two per chunk, both 64-byte halves of every line, up to 24 groups ahead (spmv_ilv_kernel; soc-LiveJournal1 shape
// 312 -> 277 us, com-Orkut shape 783 -> 704: profiles/r05_helper_wavefronts.log; an image that the cache keeps gains nothing: wiki-Talk
// shape 43.4 -> 43.7) -- and walk their workgroups backwards every other SpMV, so that what the last one streamed last is the first thing
// the next one wants (305 against 312 us).  CVR_DEBUG=ilv_helpers / ilv_ahead / ilv_per_line / ilv_flip set them for experiments.
// (launch parameters, not part of the image: cvr_create and cvr_load_image both end with this)
void ilv_runtime_settings(cvr_handle *h)
{
    size_t stream_all = 0, sums = 0;
    for (const Part &p : h->parts) if (p.img.ilv) { stream_all += p.stream_bytes; if (h->paneled()) sums += (size_t)p.yext * h->vsz; }
    // "does not stay": what an SpMV walks -- the image, the panels' partial sums, x and y -- against the 256 MiB of the cache.  (The rule was the image
    // alone above 192 MiB until the soc-LiveJournal1 shape x 0.5 -- image 181 MB, partial sums 38, x and y 39: 259 MB -- turned out to run 158 us without
    // helpers and 129 with; x 0.45 -- 232 MB -- 139.5 and 125.5; x 0.4 -- 228 MB, one generation of workgroups -- 89.3 and 90.3: profiles/r05_helper_threshold*.log.
    // The line is drawn at 230 MB: what the cache keeps of a walk is less than its 268 MB.)
    const size_t walked = stream_all + sums + (size_t)(h->info.ncols + h->info.nrows) * h->vsz;
    const bool   big = walked > (size_t)230000000;
    {   // the same question for images with a hub table (spmv_kernel, WIN == 3: cvr_spmv.hip)
        size_t hub_stream = 0, hub_sums = 0;
        for (const Part &p : h->parts) if (p.img.hub_n) { hub_stream += p.stream_bytes; if (h->paneled()) hub_sums += (size_t)p.yext * h->vsz; }
        const char *e = cvr::debug_env("ilv_stream_nt");
        for (Part &p : h->parts)
            if (p.img.hub_n && !p.img.ilv) p.img.ilv_stream_nt = e ? (uint32_t)std::max(0, atoi(e)) : hub_stream + hub_sums + (size_t)(h->info.ncols + h->info.nrows) * h->vsz > (size_t)230000000 ? 1u : 0u;
    }
    if (cvr::debug_env("fused_trace") && stream_all) fprintf(stderr, "[cvr] interleaved image: stream %.0f MB + partial sums %.0f MB + x, y %.0f MB = %.0f MB walked per SpMV: helper wavefronts %s\n", stream_all / 1e6, sums / 1e6,
                                                                  (double)(h->info.ncols + h->info.nrows) * h->vsz / 1e6, walked / 1e6, big ? "on" : "off");
    for (Part &p : h->parts) {
        if (!p.img.ilv) continue;
        const char *e;
        // (gang chunks: none -- the common list takes a CU's stream 25 % faster than four private chunks did, more of the helpers' batches come too late, and
        // what they still bring costs issue slots: soc-LiveJournal1 shape 212.0 us without, 214.2 / 220.8 with one / two per chunk; com-Orkut 594 / 589 / 600:
        // profiles/r06_gang_launch_parameters.log)
        p.img.ilv_helpers = (e = cvr::debug_env("ilv_helpers")) ? (uint32_t)std::max(0, atoi(e)) : big && !p.img.gang ? 2u : 0u;
        p.img.ilv_per_line = (e = cvr::debug_env("ilv_per_line")) ? (uint32_t)std::max(1, atoi(e)) : 2u;
        p.img.ilv_ahead = (e = cvr::debug_env("ilv_ahead")) ? (uint32_t)std::max(1, atoi(e)) : 24u;
        p.img.ilv_flip = (e = cvr::debug_env("ilv_flip")) ? (uint32_t)std::max(0, atoi(e)) : big ? 1u : 0u;
        p.img.ilv_stream_nt = (e = cvr::debug_env("ilv_stream_nt")) ? (uint32_t)std::max(0, atoi(e)) : big ? 1u : 0u;          // (and the stream past the caches: cvr_spmv.hip, ring_ld128s)
    }
    if (const char *e = cvr::debug_env("combine_batch")) { const int v = atoi(e); h->combine_batch = v == 8 || (v >= 9 && v <= 12) || v == 16 || v == 17 ? v : 4; }
    // the combine pass of a matrix whose rows are mostly empty (fewer partial sums over all panels than rows): eight blocks of rows per workgroup,
    // the loads of eight panels per round trip (combine_kernel)
    if (h->paneled()) {
        int64_t pairs = 0;
        for (const Part &p : h->parts) pairs += p.nrows;
        const bool sparse = pairs < h->info.nrows;
        const char *e = cvr::debug_env("combine_mul");
        h->combine_mul = e ? (atoi(e) == 8 ? 8 : 1) : sparse ? 8 : 1;
        // (and workgroups of 1 024 threads with one entry per thread and panel: a workgroup's eight blocks hold a few hundred sums per panel and 64 KB of y to
        // write, with about one workgroup per CU -- 256 threads left most of the CU's memory pipeline idle: wiki-Talk shape 35.5 -> 33.8 us, x 2 61.3 -> 53.7,
        // the forum-like hold-out shape 54.3 -> 52.2: profiles/r05_sparse_combine_threads.log; launch_combine: 9 = eight panels per round trip, 12 = sixteen)
        if (!cvr::debug_env("combine_batch") && sparse) h->combine_batch = h->parts.size() <= 8 ? 9 : 12;
    }
}

}  // namespace cvrh

extern "C" {

const char *cvr_last_error(void) { return g_err; }

/* diagnostics: the time stamps of the last SpMV of a handle created under CVR_DEBUG=phase_clocks (spmv_seg_kernel<.., PROF>) */
int cvr_debug_phase_clocks(cvr_handle *h, unsigned long long *out, int64_t max_words, int64_t *nwords)
{
    if (!h || !nwords) return fail(CVR_ERR_INVALID, "null argument");
    *nwords = 0;
    if (h->parts.empty() || !h->parts[0].img.prof) return fail(CVR_ERR_STATE, "the handle was not created under CVR_DEBUG=phase_clocks (or its layout has no phase clocks)");
    const int64_t n = std::min<int64_t>(max_words, h->parts[0].img.prof_words);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (out && n > 0) HIP_TRY(hipMemcpy(out, h->parts[0].img.prof, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
    *nwords = h->parts[0].img.prof_words;
    return CVR_OK;
}
const char *cvr_version(void) { return "cvr_amd 0.1 (gfx950, CVR64)"; }

void cvr_default_options(cvr_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->device = 0;
    o->steps_per_chunk = 0;
    o->split_threshold = 0;
    o->xcd_swizzle = -1;
    o->x_window = -1;
    o->col_panels = -1;
    o->value_dict = -1;
    o->col_phases = -1;
    o->hub_table = -1;
    o->narrow_cols = -1;
    o->hub_reorder = -1;
    o->row_tags16 = -1;
    o->row_bands = -1;
    o->piece_max = -1;
    o->interleave = -1;
    o->gang = -1;
}

}  // extern "C"

namespace cvrh {

int check_csr(const cvr_csr_view *c, bool columns_on_host)
{
    if (!c || c->nrows < 0 || c->ncols < 0) return fail(CVR_ERR_INVALID, "null or negative-size CSR view");
    if (c->nrows > 0 && !c->row_ptr) return fail(CVR_ERR_INVALID, "row_ptr is null");
    if (c->ncols >= (int64_t)0x7fffffff) return fail(CVR_ERR_INVALID, "ncols must be < 2^31 - 1 (bit 31 of a column word is the segment-end flag)");
    if ((uint64_t)(c->ncols + 1) * (c->is_f32 ? 4u : 8u) > 0xffffffffull)
        return fail(CVR_ERR_INVALID, "x (%lld values) exceeds the 4 GiB a buffer descriptor addresses; shard the columns", (long long)c->ncols);
    if (c->nrows == 0) return CVR_OK;
    if (c->row_ptr[0] < 0) return fail(CVR_ERR_INVALID, "row_ptr[0] < 0");
    for (int64_t r = 0; r < c->nrows; r++)
        if (c->row_ptr[r + 1] < c->row_ptr[r]) return fail(CVR_ERR_INVALID, "row_ptr decreases at row %lld", (long long)r);
    const int64_t nnz = c->row_ptr[c->nrows];
    if (nnz > 0 && (!c->col_idx || !c->vals)) return fail(CVR_ERR_INVALID, "col_idx / vals is null");
    if (!columns_on_host) return CVR_OK;       // device arrays: the range check is a kernel (check_columns_device)
    // the first offending position, searched by a few threads on large matrices (69 M columns: 17 ms on one core)
    const int64_t j0 = c->row_ptr[0];
    int           T = (int)std::thread::hardware_concurrency();
    if (T > 16) T = 16;
    if (T < 1 || nnz - j0 < (1 << 22)) T = 1;
    std::vector<int64_t> bad((size_t)T, -1);
    auto scan = [&](int t) {
        const int64_t a = j0 + (nnz - j0) * t / T, b = j0 + (nnz - j0) * (t + 1) / T;
        const int32_t nc = (int32_t)c->ncols;
        for (int64_t j = a; j < b; j++)
            if ((uint32_t)c->col_idx[j] >= (uint32_t)nc) { bad[(size_t)t] = j; return; }     // negative or >= ncols
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(scan, t);
        scan(0);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; t++)
        if (bad[(size_t)t] >= 0)
            return fail(CVR_ERR_INVALID, "col_idx[%lld] = %d outside [0, %lld)", (long long)bad[(size_t)t], c->col_idx[bad[(size_t)t]], (long long)c->ncols);
    return CVR_OK;
}

// the column range check for col_idx in device memory
// row_ptr in device memory: non-negative start, no decrease -- checked where it lies (a copy of 8 bytes per row to the host and back costs
// more than the whole analysis of a large matrix); *rp0 / *rpn = its first / last entry
int check_rows_device(const int64_t *rp_dev, int64_t nrows, int64_t *rp0, int64_t *rpn)
{
    *rp0 = 0; *rpn = 0;
    if (nrows <= 0) return CVR_OK;
    long long *d = nullptr, host[3] = {0, 0, -1};
    HIP_TRY(hipMalloc(&d, sizeof(host)));
    (void)hipGetLastError();
    hipError_t e = cvr::launch_rows_check(rp_dev, nrows, d, nullptr);
    const char *where = "launch";
    if (e == hipSuccess) { e = hipMemcpy(host, d, sizeof(host), hipMemcpyDeviceToHost); where = "copy"; }
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CVR_ERR_HIP, "row_ptr check (%s): %s", where, hipGetErrorString(e));
    if (host[0] < 0) return fail(CVR_ERR_INVALID, "row_ptr[0] < 0");
    if (host[2] >= 0) return fail(CVR_ERR_INVALID, "row_ptr decreases at row %lld", host[2]);
    *rp0 = host[0]; *rpn = host[1];
    return CVR_OK;
}
int check_columns_device(const int32_t *ci_dev, int64_t j0, int64_t j1, int64_t ncols)
{
    if (j1 <= j0) return CVR_OK;
    int32_t *mm = nullptr;
    int32_t  host[2] = {0x7fffffff, (int32_t)0x80000000};
    HIP_TRY(hipMalloc(&mm, sizeof(host)));
    hipError_t e = hipMemcpy(mm, host, sizeof(host), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = cvr::launch_col_range(ci_dev, j0, j1, mm, nullptr);
    if (e == hipSuccess) e = hipMemcpy(host, mm, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipFree(mm);
    if (e != hipSuccess) return fail(CVR_ERR_HIP, "column range check: %s", hipGetErrorString(e));
    if (host[0] < 0 || host[1] >= ncols) return fail(CVR_ERR_INVALID, "col_idx holds %d .. %d, outside [0, %lld)", host[0], host[1], (long long)ncols);
    return CVR_OK;
}

}  // namespace cvrh

extern "C" {

int cvr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int64_t cvr_plan_bound(int64_t nrows, int64_t nnz, int32_t S) { return cvr::plan_bound(nrows, nnz, S); }

int64_t cvr_plan_chunks(int64_t nrows, const int64_t *row_ptr, int32_t S, int64_t thr, int64_t *nz_begin,
                        int64_t *row_first, int64_t *nseg, int64_t *pad_cnt)
{
    if (nrows < 0 || (nrows > 0 && !row_ptr) || S < 4 || S % 4) return fail(CVR_ERR_INVALID, "bad planner arguments");
    const cvr::Plan p = cvr::plan_chunks(nrows, row_ptr, S, thr);
    const int64_t   n = (int64_t)p.chunks.size();
    for (int64_t k = 0; k < n; k++) {
        if (nz_begin) nz_begin[k] = p.chunks[k].nz_begin;
        if (row_first) row_first[k] = p.chunks[k].row_first;
        if (nseg) nseg[k] = p.chunks[k].nseg;
        if (pad_cnt) pad_cnt[k] = p.chunks[k].pad_cnt;
    }
    if (nz_begin) nz_begin[n] = p.nz_end;
    return n;
}

// diagnostics: the plan of the device planner (cvr_plan_dev.hip) against the host planner's for the same row_ptr, field by
// field; seconds of both (the device figure includes the two synchronisations, not the upload of row_ptr)
int cvr_plan_selfcheck(int device, int64_t nrows, const int64_t *row_ptr, int32_t S, int64_t thr, int64_t max_rows, double *host_s, double *device_s,
                       int64_t *nchunks)
{
    if (nrows < 0 || (nrows > 0 && !row_ptr) || S < 4 || S % 4) return fail(CVR_ERR_INVALID, "bad planner arguments");
    if (device < 0 || device >= cvr_device_count()) return fail(CVR_ERR_NO_DEVICE, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    const double    t0 = now_s();
    const cvr::Plan ph = cvr::plan_chunks(nrows, row_ptr, S, thr, max_rows);
    const double    t1 = now_s();
    if (host_s) *host_s = t1 - t0;
    if (nchunks) *nchunks = (int64_t)ph.chunks.size();
    int64_t *d_rp = nullptr;
    HIP_TRY(hipMalloc(&d_rp, sizeof(int64_t) * ((size_t)nrows + 1)));
    hipError_t e = nrows > 0 ? hipMemcpy(d_rp, row_ptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice) : hipSuccess;
    cvr::Plan        pd;
    bool             fallback = false;
    double           best = 1e30;
    cvr::PlanScratch ws;             // as cvr_create keeps it: device scratch re-used, records through a pinned buffer
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ws.pinned), ws.pinned_bytes = (size_t)640 << 10, hipHostMallocDefault);
    for (int rep = 0; rep < 3 && e == hipSuccess; rep++) {
        const double t2 = now_s();
        e = cvr::plan_chunks_device(d_rp, nrows, nrows ? row_ptr[nrows] : 0, S, thr, max_rows, &pd, &fallback, nullptr, rep ? &ws : nullptr);
        best = std::min(best, now_s() - t2);
    }
    cvr::free_plan_scratch(ws);
    (void)hipFree(d_rp);
    if (e != hipSuccess) return fail(CVR_ERR_HIP, "device planner: %s", hipGetErrorString(e));
    if (device_s) *device_s = best;
    if (fallback) return fail(CVR_ERR_STATE, "device planner declined (row block beyond 32-bit slot positions, or S too large)");
    if (pd.S != ph.S || pd.thr != ph.thr || pd.max_rows != ph.max_rows || pd.nz_end != ph.nz_end) return fail(CVR_ERR_INTERNAL, "plan parameters differ");
    if (pd.chunks.size() != ph.chunks.size()) return fail(CVR_ERR_INTERNAL, "device plan has %zu chunks, host plan %zu", pd.chunks.size(), ph.chunks.size());
    for (size_t k = 0; k < ph.chunks.size(); k++) {
        const cvr::Chunk &a = ph.chunks[k], &b = pd.chunks[k];
        if (a.nz_begin != b.nz_begin || a.row_first != b.row_first || a.nrows_in != b.nrows_in || a.nseg != b.nseg || a.pad_cnt != b.pad_cnt ||
            a.head_shared != b.head_shared || a.tail_shared != b.tail_shared)
            return fail(CVR_ERR_INTERNAL, "chunk %zu differs: host {nz %lld row %lld rows %lld seg %lld pad %lld %d%d} device {nz %lld row %lld rows %lld seg %lld pad %lld %d%d}", k,
                        (long long)a.nz_begin, (long long)a.row_first, (long long)a.nrows_in, (long long)a.nseg, (long long)a.pad_cnt, a.head_shared, a.tail_shared,
                        (long long)b.nz_begin, (long long)b.row_first, (long long)b.nrows_in, (long long)b.nseg, (long long)b.pad_cnt, b.head_shared, b.tail_shared);
    }
    if (pd.shared.size() != ph.shared.size()) return fail(CVR_ERR_INTERNAL, "device plan has %zu cut rows, host plan %zu", pd.shared.size(), ph.shared.size());
    for (size_t k = 0; k < ph.shared.size(); k++)
        if (pd.shared[k].row != ph.shared[k].row || pd.shared[k].c0 != ph.shared[k].c0 || pd.shared[k].c1 != ph.shared[k].c1)
            return fail(CVR_ERR_INTERNAL, "cut row %zu differs: host {%lld %lld %lld} device {%lld %lld %lld}", k, (long long)ph.shared[k].row, (long long)ph.shared[k].c0,
                        (long long)ph.shared[k].c1, (long long)pd.shared[k].row, (long long)pd.shared[k].c0, (long long)pd.shared[k].c1);
    return CVR_OK;
}

}  // extern "C"

extern "C" {

namespace {
// CVR_CREATE_TIMING=1: wall time of the phases of cvr_create on stderr (diagnostics only)
struct PhaseClock {
    bool   on = cvr::debug_env("create_timing") && atoi(cvr::debug_env("create_timing"));
    double t = now_s();
    void   lap(const char *what)
    {
        if (cvr::debug_env("sticky")) { const hipError_t pe = hipPeekAtLastError(); if (pe != hipSuccess) fprintf(stderr, "[cvr_create] sticky error at \"%s\": %s\n", what, hipGetErrorString(pe)); }
        if (on) { const double n = now_s(); fprintf(stderr, "[cvr_create] %-28s %8.2f ms\n", what, (n - t) * 1e3); t = n; }
    }
};
}  // namespace

int cvr_create(cvr_handle **out, const cvr_csr_view *csr_in, const cvr_options *opt_in)
{
    cvr::debug_refresh();          // (first: everything below, the phase clock included, reads this call's CVR_DEBUG)
    PhaseClock clk;
    if (!out) return fail(CVR_ERR_INVALID, "out is null");
    *out = nullptr;
    Range range("cvr_create (validate, plan, upload)");
    if (!csr_in) return fail(CVR_ERR_INVALID, "null or negative-size CSR view");
    IOpt opt = make_iopt(opt_in);
    const int ndev = cvr_device_count();
    const bool on_device = csr_in->arrays_on_device != 0;
    int rc = on_device ? CVR_OK : check_csr(csr_in);          // host arrays: rejected before any device work
    if (rc) return rc;
    clk.lap("options, check_csr (host)");
    if (ndev <= 0) return fail(CVR_ERR_NO_DEVICE, "no HIP device visible: libcvr_amd has no CPU fallback");
    if (opt.device < 0 || opt.device >= ndev) return fail(CVR_ERR_NO_DEVICE, "device %d out of range [0, %d)", opt.device, ndev);
    { const Chip chip = chip_of(opt.device); opt.cus = chip.cus; opt.xcds = chip.xcds; }
    if (opt.xcds != 8 && opt.xcd_swizzle != 0) opt.xcd_swizzle = 0;      // (the chunk-range-per-XCD mapping is written for the whole chip's eight)
    if (opt.row_bands > 1) return fail(CVR_ERR_INVALID, "row_bands is reserved: leave it at its default");
    if (opt.steps_per_chunk != 0 && (opt.steps_per_chunk < 4 || opt.steps_per_chunk % 4 || opt.steps_per_chunk > 4096))
        return fail(CVR_ERR_INVALID, "steps_per_chunk must be a multiple of 4 in [4, 4096]");
    // (interleaved chunks are sorted by one workgroup in LDS: refused here, before any planning or allocation -- the converter would fail with
    // hipErrorInvalidValue after all of it)
    if (opt.interleave > 0 && opt.steps_per_chunk > cvr::kIlvMaxSteps)
        return fail(CVR_ERR_INVALID, "interleave = 1 takes steps_per_chunk <= %d (a chunk is sorted in one workgroup's LDS)", cvr::kIlvMaxSteps);

    // CSR arrays already in device memory (of opt.device): the row_ptr of a small matrix comes back once for the argument checks (8 B per
    // row), that of a large one is checked by a kernel and stays (R-MAT-24: 33 ms of copies -> 0.5 ms);
    // col_idx and vals stay where they are and are copied device to device; the chunk plan (from 200 000 rows on), the panel
    // rule and the panel split run on the device arrays (cvr_plan_dev.hip, cvr_split.hip).
    cvr_csr_view          hostv = *csr_in;
    const cvr_csr_view   *csr = &hostv;
    std::vector<int64_t>  rp_host;
    bool                  rows_on_device = false;        // device arrays of a large matrix: row_ptr never comes to the host
    int64_t               dev_j0 = 0, dev_j1 = 0;        // (then: its first and last entry)
    hipMemcpyKind         civa_kind = hipMemcpyHostToDevice;
    if (on_device) {
        if (hostv.nrows < 0 || hostv.ncols < 0 || (hostv.nrows > 0 && !hostv.row_ptr)) return fail(CVR_ERR_INVALID, "null or negative-size CSR view");
        HIP_TRY(hipSetDevice(opt.device));
        int64_t j0 = 0, j1 = 0;
        if (hostv.nrows >= device_plan_rows() && hostv.nrows > 0 && !cvr::debug_env("device_rows_to_host")) {
            // a matrix the device plans anyway: its row pointers are checked and used where they are (rows_on_device)
            cvr_csr_view shape = hostv;
            shape.nrows = 0; shape.row_ptr = nullptr;                 // (the checks of check_csr that need no rows: sizes, ncols, x within 4 GiB)
            rc = check_csr(&shape, false);
            if (rc) return rc;
            rc = check_rows_device(hostv.row_ptr, hostv.nrows, &j0, &j1);
            if (rc) return rc;
            if (j1 > j0 && (!hostv.col_idx || !hostv.vals)) return fail(CVR_ERR_INVALID, "col_idx / vals is null");
            rows_on_device = true;
            hostv.row_ptr = nullptr;                                   // (nothing below may read rows on the host)
        } else {
            rp_host.resize((size_t)hostv.nrows + 1, 0);
            if (hostv.nrows > 0) HIP_TRY(hipMemcpy(rp_host.data(), hostv.row_ptr, sizeof(int64_t) * rp_host.size(), hipMemcpyDeviceToHost));
            hostv.row_ptr = rp_host.data();
            rc = check_csr(&hostv, false);
            if (rc) return rc;
            j0 = rp_host.front(); j1 = rp_host.back();
        }
        dev_j0 = j0; dev_j1 = j1;
        rc = check_columns_device(hostv.col_idx, j0, j1, hostv.ncols);
        if (rc) return rc;
        const double xb = (double)hostv.ncols * (hostv.is_f32 ? 4.0 : 8.0);
        civa_kind = hipMemcpyDeviceToDevice;
        if (opt.col_panels < 0 && !(j1 > 0 && (xb >= kNoWindowPanelBytes || (xb >= kMidPanelBytes && resident_out_of_reach(hostv.nrows, j1 - j0, hostv.ncols, hostv.is_f32 != 0, opt)))))
            opt.col_panels = 1;      // (else: the panel rule runs on the device arrays below -- from 8 MB of x on it may be asked after the layout probe: ask_window)
        // (column panels of device arrays are split on the device: cvr_split.hip)
    }

    const int64_t nrows = csr->nrows, ncols = csr->ncols;
    const bool    f32 = csr->is_f32 != 0;
    const size_t  vsz = f32 ? 4 : 8;
    // column panels: asked for, or (col_panels < 0: auto) when x is several times the 4-MiB L2 of an XCD AND a sizeable
    // share of the gathers would miss such an L2 (l2_miss_estimate: a banded matrix re-uses its few lines, R-MAT's hub
    // columns stay resident).  One panel per 1.8 MB of *missing* x: LiveJournal shape (38.8 MB, 0.44) -> 9, the best of
    // 4..32 (profiles/r01_column_panels_livejournal_sweep2.log); R-MAT-24 fp32 (67 MB, 0.22) -> 8, 1 660 against
    // 2 300 us as one image (profiles/r01_column_panels_rmat24_fp32.log); R-MAT-22 fp64 (33.5 MB, 0.13) and matrices
    // whose x nearly fits (web-Google: profiles/r01_column_panel_probe.log) stay whole.
    const bool panels_auto = !(opt_in && opt_in->col_panels >= 0);
    clk.lap("device arrays: row_ptr, checks");
    if (nrows >= (int64_t)0xfffffff0u) return fail(CVR_ERR_INVALID, "matrix too large for 32-bit row ordinals on one GPU");

    cvr_handle *h = new (std::nothrow) cvr_handle;
    if (!h) return fail(CVR_ERR_NOMEM, "out of host memory");
    h->device = opt.device;
    h->vsz = vsz;
    h->opt_used = make_iopt(opt_in);
    h->opt_used.cus = opt.cus; h->opt_used.xcds = opt.xcds;
    cvr_info &in = h->info;
    in.nrows = nrows; in.ncols = ncols; in.nnz = rows_on_device ? dev_j1 - dev_j0 : nrows ? csr->row_ptr[nrows] - csr->row_ptr[0] : 0; in.is_f32 = f32 ? 1 : 0;
    in.x_elems = ncols + 1;
#define CREATE_TRY(expr)                                                                                    \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);          \
            cvr_destroy(h);                                                                                 \
            return CVR_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)
    CREATE_TRY(hipSetDevice(h->device));
    // The first cvr_create of a process: the runtime loads a file's code object when one of its kernels is first asked for (milliseconds, on the
    // calling thread).  A thread of its own asks now, beside the allocations and the upload below; joined before this call returns.
    struct Warm {
        std::thread t;
        ~Warm() { if (t.joinable()) t.join(); }
    } warm;
    {
        static std::atomic<bool> warmed{false};
        if (!warmed.exchange(true) && !cvr::debug_env("no_warm_thread"))
            warm.t = std::thread([dev = h->device] {
                if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return; }
                cvr::touch_plan_kernels(); cvr::touch_convert_kernels(); cvr::touch_spmv_kernels();
                (void)side_stream(dev, 0); (void)side_stream(dev, 1);      // (created on first use otherwise: milliseconds each)
            });
    }
    CREATE_TRY(acquire_stream(h->device, &h->stream));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->plan_ws.pinned), h->plan_ws.pinned_bytes = nrows >= device_plan_rows() ? (size_t)704 << 10 : kPinnedSmall, hipHostMallocDefault));
    CREATE_TRY(hipMalloc(&h->d_small, kSmallBytes));
    // the small scratch starts out as its first users want it (probe output and flags zero, dictionary table all ones), and the
    // two events of cvr_preprocess / cvr_spmv_bench exist: none of that in the timed analysis
    CREATE_TRY(hipMemsetAsync(h->d_small, 0, kSmallBytes, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_small + kSmallDictTab, 0xff, sizeof(unsigned long long) * 1024, h->stream));
    h->small_clean = true;
    if (nrows >= device_plan_rows()) {     // the device planner's scratch, sized for chunks of 16 steps or more (it grows if the plan needs more)
        const int64_t nnz0 = rows_on_device ? dev_j1 : nrows ? csr->row_ptr[nrows] : 0, nb = nrows / cvr::kPlanRowBlock + 1;
        const size_t  want = (size_t)nrows * 6 + (size_t)nb * 16 + (size_t)nrows / 256 + (size_t)(2 * ((nnz0 + nrows) / 1024) + 3 * nb) * 96 + 8192;
        if (hipMalloc(&h->plan_ws.dev, want) == hipSuccess) h->plan_ws.dev_bytes = want; else (void)hipGetLastError();
    }
    h->events.resize(2);
    CREATE_TRY(hipEventCreate(&h->events[0]));
    CREATE_TRY(hipEventCreate(&h->events[1]));
    clk.lap("handle, stream");
    const double t_up0 = now_s();
    // Host arrays of a matrix that may get column panels (x of 24 MB or more -- 12 MB beyond the resident layout --, or panels asked for) are uploaded once, as they
    // are: the panel rule and the split run on that copy (building split arrays on the host means allocating, touching and
    // freeing another copy of the matrix there, which costs more than the PCIe transfer: LiveJournal shape 60 ms to split +
    // 130 ms to free against 20 ms to upload), and a matrix that stays whole adopts it as its device CSR.  The host split
    // stays as the fallback for matrices beyond the device split's 32-bit positions or when the copy does not fit.
    struct Staged {          // (one allocation: every hipFree of a large buffer takes ~190 us with the GPU idle behind it)
        void *rp = nullptr, *ci = nullptr, *va = nullptr;
        void  release() { (void)hipFree(rp); rp = ci = va = nullptr; }
        ~Staged() { release(); }
    } staged;
    const int64_t  sj0 = rows_on_device ? dev_j0 : nrows ? csr->row_ptr[0] : 0, sj1 = rows_on_device ? dev_j1 : nrows ? csr->row_ptr[nrows] : 0;
    const int64_t *rp_d = csr_in->row_ptr;
    const int32_t *ci_d = csr_in->col_idx;
    const void    *va_d = csr_in->vals;
    bool           dev_split = on_device;
    int            P = opt.col_panels;
    const double   xbytes = (double)ncols * (double)vsz;
    // (x of 12 .. 24 MB: the rule also runs for matrices too large for the resident layout -- web-Google shapes of 12-16 M non-zeros run
    // 21-28 % faster as eight panels, one per XCD, than as one plain image: profiles/r03_mid_size_panels.log)
    bool           mid_range = P < 0 && xbytes >= kMidPanelBytes && xbytes < 24e6 && sj1 > sj0 && resident_out_of_reach(nrows, sj1 - sj0, ncols, f32, opt);
    // (x of 8 .. 24 MB with every layout option left to the rules: the rule also runs for matrices that have no use for the resident layout's window -- decided below, on
    // the device copy, by the layout probe's near-diagonal share: wiki-Talk-like matrices of 9.6 / 14.4 MB of x ran 37.4 / 60.8 us as one image, 21.4 / 23.9 as sixteen
    // gang panels: profiles/r06_thin_lists_rule.log)
    const bool     ask_window = P < 0 && !mid_range && xbytes >= kNoWindowPanelBytes && xbytes < 24e6 && sj1 > sj0 && nrows >= 4096 && ncols >= 4096 && opt.steps_per_chunk == 0 && opt.waves_per_block == 0 &&
                                opt.x_window < 0 && opt.col_phases < 0 && opt.interleave < 0 && opt.gang < 0 && opt.hub_table < 0 && !opt.debug_col_mask && !cvr::debug_env("no_auto_layout") && !cvr::debug_env("no_window_question");
    if (!on_device && (P > 1 || (P < 0 && (xbytes >= 24e6 || mid_range || ask_window))) && sj1 > 0 && sj1 < (int64_t)0xffffffffll && !cvr::debug_env("host_split")) {
        auto         up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t b_rp = up(sizeof(int64_t) * ((size_t)nrows + 1)), b_ci = up(sizeof(int32_t) * (size_t)sj1), b_va = up(vsz * (size_t)sj1);
        if (hipMalloc(&staged.rp, b_rp + b_ci + b_va) == hipSuccess) {
            staged.ci = static_cast<uint8_t *>(staged.rp) + b_rp;
            staged.va = static_cast<uint8_t *>(staged.ci) + b_ci;
        }
        if (staged.rp &&
            hipMemcpy(staged.rp, csr->row_ptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(staged.ci, csr->col_idx, sizeof(int32_t) * (size_t)sj1, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(staged.va, csr->vals, vsz * (size_t)sj1, hipMemcpyHostToDevice) == hipSuccess) {
            rp_d = static_cast<const int64_t *>(staged.rp); ci_d = static_cast<const int32_t *>(staged.ci); va_d = staged.va;
            dev_split = true;
        } else {
            (void)hipGetLastError();      // not enough device memory for the staging copy: rule and split on the host
            staged.release();
        }
    }
    clk.lap("staging upload");
    bool no_window = false;
    if (ask_window && dev_split) {          // the layout probe in front of the panel rule: the share of the non-zeros the resident layout's window would hold
        unsigned long long *d_out = reinterpret_cast<unsigned long long *>(h->d_small + kSmallProbe);
        std::vector<unsigned long long> outv(2 * cvr::kProbeBlocks, 0);
        const int64_t win = (96 * 1024) / (int64_t)vsz;
        hipError_t e = cvr::launch_probe(rp_d, ci_d, nrows, ncols, (uint32_t)(win / 4), d_out, h->stream, h->small_clean);
        if (e == hipSuccess) e = hipMemcpyAsync(outv.data(), d_out, sizeof(unsigned long long) * outv.size(), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * outv.size(), h->stream);          // (the probe's part of the small scratch is clean again: auto_layout, the early dictionary scan)
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { cvr_destroy(h); return fail(CVR_ERR_HIP, "layout probe: %s", hipGetErrorString(e)); }
        unsigned long long cnt = 0;
        for (uint32_t b = 0; b < cvr::kProbeBlocks; b++) cnt += outv[2 * b + 1];
        const double near = (double)cnt / std::max<double>((double)(sj1 - sj0), 1.0);
        if (near < 0.15) { mid_range = true; no_window = true; }          // (no window for it: the panel question is asked as for a matrix beyond the resident layout)
        if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] x %.1f MB, %.3f of the non-zeros near the diagonal: the panel question is %s\n", xbytes / 1e6, near, mid_range ? "asked" : "not asked (resident layout)");
    }
    // the panel rule (col_panels < 0): on the device copy when there is one (same windows, same integers as the host form)
    const double t_rule0 = now_s();
    double       rule_miss = -1;          // >= 0: the device rule ran and its count is still to be weighed against the partial sums it costs (below)
    if (P < 0) {
        if (!(xbytes >= 24e6 || mid_range) || sj1 <= sj0) P = 1;
        else if (dev_split) {
            double miss = 0;
            CREATE_TRY(l2_miss_estimate_dev(rp_d, ci_d, nrows, ncols, f32, h->stream, &miss, cvr::Scratch{h->plan_ws.dev, h->plan_ws.dev_bytes}));      // (the planner's scratch is idle until the split is done)
            P = panels_from_miss(xbytes, miss);
            rule_miss = miss;
            if (cvr::debug_env("panel_rule_trace")) fprintf(stderr, "[cvr] panel rule: L2 miss share %.17g of the gathers in eight windows of rows -> %d panels\n", miss, P);
        } else P = auto_panels(*csr, nullptr);          // (the host rule asks both questions itself)
    }
    const double panel_rule_s = now_s() - t_rule0;          // part of the analysis: added to plan_s below
    clk.lap("panel rule");
    // panels in rounds of eight, each on one XCD (run_spmv, d_multi; cvr_panels.hip: xcd_panel_count)
    const char *xp_env = cvr::debug_env("xcd_panels");
    const bool  xcd_panels = !(xp_env && atoi(xp_env) == 0) && opt.xcds == 8;
    if (P > 1 && panels_auto && xcd_panels && dev_split) P = xcd_panel_count(P, xbytes);      // (the host rule, auto_panels, has counted them that way already)
    // Power-law matrices whose popular columns will sit in hub tables need fewer, wider panels: the table takes the hot
    // half of the gathers off the L2s, and what remains runs best with ~16 MB of x per panel instead of ~4 (R-MAT-26 fp32
    // on one GPU: 59 panels 6.4 ms, 16 panels 5.4 ms; R-MAT-24: 8 and 4 panels alike; profiles/r02_hub_table_rmat.log)
    if (P > 1 && dev_split && panels_auto && opt.hub_table < 0 && opt.waves_per_block == 0 && opt.x_window <= 0 && !cvr::debug_env("no_auto_layout")) {      // (the predicate of choose_hubs: only panels that will get tables are widened)
        const int64_t room = ((int64_t)cvr::kLdsBytes / (int64_t)vsz - 8 * (cvr::kLanes + 512) - cvr::kDictMax - 8) & ~(int64_t)1023;
        double            share = 0;
        std::vector<double> col_share(cvr::kColBins, 0.0);          // the non-zeros' shares of 1 024 equal column ranges (same pass)
        const double      th0 = now_s();
        const hipError_t  e = cvr::hub_share_device(ci_d, sj0, sj1, ncols, (uint32_t)std::max<int64_t>(room, 1024), &share, h->stream, cvr::Scratch{h->plan_ws.dev, h->plan_ws.dev_bytes}, col_share.data());      // (the share alone: no ranking of the columns)
        in.hub_select_s += now_s() - th0;
        if (e != hipSuccess) { cvr_destroy(h); return fail(CVR_ERR_HIP, "hub selection: %s", hipGetErrorString(e)); }
        if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] popularity: the %lld most popular columns hold %.3f of the non-zeros (%d panels so far, x %.1f MB)\n", (long long)std::max<int64_t>(room, 1024), share, P, xbytes / 1e6);
        if (share >= 0.25) {       // (of the whole matrix: the panels' own tables, ranked inside their ranges, hold more)
            const int Pw = std::max(2, (int)std::ceil((double)ncols * (double)vsz / 16e6));
            if (mid_range) P = 1;      // (an x of one such panel: the single image with its hub table, as before -- R-MAT-22 fp32)
            else if (Pw < P) P = Pw;
        }
        // popularity too flat for the panels to be widened for their tables (a panel's own top columns hold a few times the whole
        // matrix's share at most: LiveJournal shape 0.06): the panels skip their own counting passes and take interleaved chunks instead
        // (web-Google shape x 2.2, whose share lies between 0.08 and 0.25: 72 us as plain panels that end up without tables, 55 us interleaved)
        if (share < (cvr::debug_env("flat_share") ? atof(cvr::debug_env("flat_share")) : 0.25)) opt.hub_table = 0;
        // Panels that run one per XCD are equally WIDE, not equally full: when the non-zeros crowd into some column ranges, the XCD with the
        // fullest panels sets the time (round 5 hold-out: a bipartite matrix whose columns thin out towards the end -- the first of 16 panels
        // holds 15.7 % of the non-zeros, 1.26 x an XCD's fair share -- ran 187 us as 16 panels, 112 us as 32).  The shares of the panels come
        // from the pass above; the count is doubled (up to 64) until the heaviest XCD -- panels dealt as run_spmv's table deals them, the
        // fullest first to the least loaded XCD -- is within 30 % of the mean (15 % until round 6: with gang chunks the same bipartite shape runs 78.4 / 79.3 / 83.0 us as
        // 8 / 16 / 32 panels -- its 16 panels at 1.21 of the mean no longer want doubling; profiles/r06_holdout.log).
        // (Round 6: for PRIVATE interleaved chunks only.  With gang chunks the same shape runs 78.4 / 79.3 / 83.0 us as 8 / 16 / 32 panels and, at 0.45 of its size,
        // 54.2 as 8 panels at 1.53 of the mean against 65.0 as 16 at 1.21 -- a sparse gang's time goes with its rows, not with its non-zeros: profiles/r06_thin_lists_rule.log)
        if (opt.hub_table == 0 && xcd_panels && P > 1 && P <= 64 && (opt.gang == 0 || cvr::debug_env("balance_rule")) && !cvr::debug_env("no_balance_rule")) {
            auto imbalance = [&](int Pt) {
                std::vector<double> load((size_t)Pt, 0.0);
                const int64_t       w = (ncols + Pt - 1) / Pt > 0 ? (ncols + Pt - 1) / Pt : 1;
                for (uint32_t b = 0; b < cvr::kColBins; b++) load[(size_t)std::min<int64_t>(Pt - 1, (int64_t)(((double)b + 0.5) * (double)ncols / cvr::kColBins) / w)] += col_share[b];
                std::sort(load.begin(), load.end(), std::greater<double>());
                const size_t rounds = ((size_t)Pt + 7) / 8;
                double       x[8] = {0, 0, 0, 0, 0, 0, 0, 0}, all = 0;
                size_t       used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (double l : load) {
                    int best = -1;
                    for (int q = 0; q < 8; q++) if (used[q] < rounds && (best < 0 || x[q] < x[best])) best = q;
                    x[best] += l; used[best]++; all += l;
                }
                return all > 0 ? *std::max_element(x, x + 8) * 8.0 / all : 1.0;
            };
            int    Pb = P;
            double ib = imbalance(P);
            for (int Pt = 2 * P; ib > 1.30 && Pt <= 64; Pt *= 2) { const double it = imbalance(Pt); if (it < ib - 0.02) { ib = it; Pb = Pt; } }
            if (cvr::debug_env("fused_trace") && Pb != P) fprintf(stderr, "[cvr] %d column panels instead of %d: the heaviest XCD at %.2f of the mean instead of %.2f\n", Pb, P, ib, imbalance(P));
            P = Pb;
        }
    }
    // the rule's second question, for the count it arrived at: do the panels' partial sums cost less than the misses they save?  (cvr_panels.hip: panels_pay)
    if (P > 1 && panels_auto && dev_split && rule_miss >= 0 && !cvr::debug_env("no_pairs_rule")) {
        double ppn = 0;
        const double tq = now_s();
        CREATE_TRY(pairs_per_nnz_dev(rp_d, ci_d, nrows, (ncols + P - 1) / P > 0 ? (ncols + P - 1) / P : 1, h->stream, &ppn, cvr::Scratch{h->plan_ws.dev, h->plan_ws.dev_bytes}));
        if (!panels_pay(rule_miss, ppn)) {
            if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] %d column panels dropped: L2 miss share %.3f against %.3f (row, panel) pairs per non-zero\n", P, rule_miss, ppn);
            P = 1;
            opt.hub_table = opt_in ? opt_in->hub_table : -1;          // (the flat-popularity shortcut above was taken for panels)
            // The gathers of such a matrix do miss the L2s (the rule asked for panels), only panels would write more partial sums than they save misses: short rows
            // whose columns are spread, but not evenly -- the citation-like hold-out shape.  What is left to save is the NUMBER of requests: one interleaved image
            // with gang chunks (one sorted list of ~100 000 non-zeros per workgroup) -- 134 us plain, 123 interleaved, 111 as gangs (profiles/r06_holdout_single_image.log);
            // every other hold-out shape that keeps one image has no misses to speak of (road-like) or a popular head (R-MAT: hub table) and never comes here
            // (not rows of one or two non-zeros: nothing to accumulate per row, and the accumulators' row cap would end their chunks early)
            if (sj1 - sj0 >= 3 * nrows && opt.interleave < 0 && opt.gang < 0 && opt.hub_table < 0 && opt.steps_per_chunk == 0 && opt.waves_per_block == 0 && opt.x_window < 0 && opt.col_phases < 0 && !cvr::debug_env("no_auto_layout") && !cvr::debug_env("no_single_gang")) {
                opt.interleave = 1; opt.gang = 1; opt.hub_table = 0;
            }
        }
        // Thin lists (round 6, late): a gang's sorted list shares lines of x once it holds a few non-zeros per line of its panel's slice -- the requests per non-zero
        // follow (1 - exp(-d)) / d for d = non-zeros of a gang / lines of the slice (measured 0.64 / 0.227 / 0.215 at d = 1.1 / 4.9 / 4.8: wiki-Talk, soc-LiveJournal1,
        // com-Orkut shapes).  A matrix too small to give its gangs two non-zeros per line gets panels half as wide, when the partial sums that costs are few:
        // wiki-Talk shape (d = 1.1, 0.05 pairs per non-zero) 33.2 us as 8 panels, 28.7 as 16; not the forum-like shape (d = 1.5, but 0.32 pairs per non-zero: 45.5 / 49.0)
        // nor anything with d >= 2 (wiki-Talk x 2: 46.9 as 16, 49.5 as 32) -- profiles/r06_thin_lists_rule.log
        if (P > 1 && P <= 32 && xcd_panels && opt.hub_table == 0 && opt.interleave != 0 && opt.gang != 0 && opt.waves_per_block == 0 && !cvr::debug_env("no_auto_layout") && !cvr::debug_env("no_thin_lists_rule")) {
            if (thin_lists(P, xbytes, sj1 - sj0, opt.cus, ppn)) {          // (cvr_panels.hip)
                if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] %d column panels instead of %d: fewer than two non-zeros of a gang per line of a panel's slice of x, %.3f (row, panel) pairs per non-zero\n", 2 * P, P, ppn);
                P *= 2;
            }
        }
        in.plan_s += now_s() - tq;
    }
    clk.lap("panel count with hub tables");
    if (P < 1) P = 1;
    if (P > 64) P = 64;
    in.col_panels = P;
    h->parts.resize((size_t)P);
    if (P == 1 && opt.interleave < 0) opt.interleave = 0;      // (a single image is interleaved only when asked to)
    if (P == 1 && opt.gang < 0) opt.gang = 0;                  // (and its chunks ganged only when asked to)
    if (P == 1) {
        if (staged.rp) {          // the staging copy becomes the part's device CSR
            Part &part = h->parts[0];
            part.d_rp = static_cast<int64_t *>(staged.rp); part.d_ci = static_cast<int32_t *>(staged.ci); part.d_va = staged.va;
            part.csr_borrowed = true;          // (one allocation: d_rp is its base and the part's to free, d_ci / d_va lie inside it)
            staged.rp = staged.ci = staged.va = nullptr;
        }
        if (rows_on_device) {       // the part's row pointers: a device copy of the caller's; planned (and, where the layout allows, converted) from there
            Part &part = h->parts[0];
            CREATE_TRY(hipMalloc(&part.d_rp, sizeof(int64_t) * ((size_t)nrows + 1)));
            CREATE_TRY(hipMemcpyAsync(part.d_rp, csr_in->row_ptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyDeviceToDevice, h->stream));
            const DevRows dr{part.d_rp, sj0, sj1, h->stream, &h->plan_ws};
            rc = build_part(h, part, nrows, ncols, nullptr, csr->col_idx, csr->vals, civa_kind, f32, opt, &in.plan_s, nullptr, &dr);
        } else
            rc = build_part(h, h->parts[0], nrows, ncols, csr->row_ptr, csr->col_idx, csr->vals, civa_kind, f32, opt, &in.plan_s);
        if (rc) { cvr_destroy(h); return rc; }
        in.yext_elems = h->parts[0].yext;
    } else {
        PanelSplit   sp;
        struct SplitGuard { cvr::DeviceSplit d; ~SplitGuard() { cvr::free_device_split(d); } } dsg;
        const double t0 = now_s();
        std::vector<int64_t> nsubs((size_t)P, 0);
        if (dev_split) {        // nothing but a few counts comes to the host: the sub-rows are planned where they are (cvr_plan_dev.hip)
            const int64_t  width = (ncols + P - 1) / P > 0 ? (ncols + P - 1) / P : 1;
            // the distinct values are looked for in the unsplit array on a second stream, beside the split (round 5: 0.3 ms of the soc-LiveJournal1
            // shape's preprocessing stood behind the combine tables as a pass of its own over the sixteen panels)
            hipStream_t dict_stream = nullptr;
            if (opt.value_dict != 0 && sj1 > sj0 && !h->preconverted && !h->dict_scanned && h->small_clean && !cvr::debug_env("no_early_dict")) {
                dict_stream = side_stream(h->device, 0);
                if (dict_stream == h->stream) dict_stream = nullptr;
                if (dict_stream && hipStreamSynchronize(h->stream) != hipSuccess) { (void)hipGetLastError(); dict_stream = nullptr; }      // (the table's initial fill was enqueued on the handle's stream: with panels asked for, nothing has waited for it yet)
                if (dict_stream) {
                    h->dict_tab.assign(1024, ~0ull);
                    if (enqueue_dict_scan(h, va_d, sj0, sj1, f32, true, h->dict_tab.data(), h->dict_flags, true, dict_stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(dict_stream); dict_stream = nullptr; }
                }
            }
            const hipError_t e = cvr::split_panels_device(rp_d, ci_d, va_d, f32, nrows, sj0, sj1, width, P, &dsg.d, h->stream);
            if (dict_stream) {       // (before the staging copy goes)
                const hipError_t ed = hipStreamSynchronize(dict_stream);
                h->small_clean = false;
                if (ed == hipSuccess) h->dict_scanned = true; else (void)hipGetLastError();      // (the pass over the panels below runs instead)
            }
            if (e != hipSuccess) { cvr_destroy(h); return fail(CVR_ERR_HIP, "column-panel split on the device: %s", hipGetErrorString(e)); }
            staged.release();       // the split arrays replace the staging copy
            for (int p = 0; p < P; p++) nsubs[(size_t)p] = dsg.d.sub0[p + 1] - dsg.d.sub0[p];
        } else {
            split_panels(*csr, P, sp);
            for (int p = 0; p < P; p++) nsubs[(size_t)p] = (int64_t)sp.rows[(size_t)p].size();
        }
        in.plan_s += now_s() - t0;      // (the split; the hub count that sizes the panels ran before t0 and is reported on its own)
        clk.lap("panel split");
        std::vector<PartPlan> pps((size_t)P);
        IOpt                  panel_opt = opt;
        panel_opt.col_phases = cvr::debug_env("panel_phases") ? atoi(cvr::debug_env("panel_phases")) : 1;          // column phases are for the single image whose chunks are all resident at once
        panel_opt.panel_on_one_xcd = xcd_panels ? 1 : 0;
        // interleaved chunks (automatic): panels that run one per XCD (their slice of x stays in that L2: what is left to save is the
        // number of requests) and get no hub tables -- scattered columns without a popular head, the soc-LiveJournal1 shape.  From 4 M
        // non-zeros on (with the chunk length that fills whole generations of workgroups: wiki-Talk shape, 5 M, 48.6 -> 43.6 us; a 2.4-M-row
        // matrix of single-entry rows 36.9 -> 38.6: stays plain; web-Google shape x 2.2 / 2.6 / 3: 72 -> 55, 84 -> 65, 95 -> 86; com-Orkut
        // shape 1 307 -> 765: profiles/r04_ilv_auto_probe.log)
        if (panel_opt.interleave < 0) panel_opt.interleave = xcd_panels && dev_split && panel_opt.hub_table == 0 && panel_opt.waves_per_block == 0 && panel_opt.x_window <= 0 && sj1 - sj0 >= (cvr::debug_env("ilv_min_nnz") ? atoll(cvr::debug_env("ilv_min_nnz")) : no_window ? (int64_t)1 << 20 : (int64_t)4 << 20) && !cvr::debug_env("no_auto_layout") ? 1 : 0;
        int ilv_generations = 0;          // > 0: the chunk length was chosen for this many generations of workgroups (checked against the plan below)
        if (panel_opt.interleave > 0) {
            panel_opt.hub_table = 0; panel_opt.col_phases = 1;
            int64_t nsub_all = 0;
            for (int64_t v : nsubs) nsub_all += v;
            // Mostly-empty matrices -- fewer (row, panel) pairs than rows: the wiki-Talk shape, 0.27 per row -- get two wavefronts per workgroup
            // instead of four: their chunks are short (the row cap of the LDS accumulators ends them, not the steps), and with two wavefronts a
            // chunk may span twice the rows (wiki-Talk shape 39.7 -> 36.6 us, x 2 at another seed 70.3 -> 66.0; every other shape of
            // profiles/r05_wpb_probe.log loses 1-20 % with two, which is why this is not the general rule)
            // gang chunks (automatic): the four chunks of an interleaved workgroup sorted together and walked by its wavefronts in turn -- four times the
            // non-zeros share the lines of x a gather touches (prototype: soc-LiveJournal1 shape 246 -> 178 us, com-Orkut shape 736 -> 510: profiles/r06_token_probe_*.log)
            if (panel_opt.gang < 0) panel_opt.gang = panel_opt.waves_per_block == 0 && !cvr::debug_env("no_gang") ? 1 : 0;
            // (round 5's rule of two wavefronts for mostly-empty matrices holds only for private chunks: with a gang the four chunks' rows share one list anyway --
            // forum-like hold-out shape 53.4 us with two wavefronts, 48.6 as a gang of four; wiki-Talk x 2: 54.2 / 50.2; wiki-Talk itself 35.8 / 36.1: profiles/r06_holdout.log)
            if (panel_opt.waves_per_block == 0 && panel_opt.gang <= 0 && panels_auto && nsub_all < nrows && !cvr::debug_env("no_sparse_waves")) panel_opt.waves_per_block = 2;
            if (panel_opt.steps_per_chunk == 0) {       // one chunk length for all panels (they share a launch): from the mean sub-row and the mean panel
                IOpt one = panel_opt;
                panel_opt.steps_per_chunk = interleave_steps((sj1 - sj0) / P, std::max<int64_t>(nsub_all / P, 1), f32, one);
                if (xcd_panels && dev_split && !cvr::debug_env("ilv_plain_steps")) {      // panels one per XCD: the length that fills whole generations of workgroups
                    std::vector<int64_t> pnz((size_t)P);
                    for (int p = 0; p < P; p++) pnz[(size_t)p] = dsg.d.off[p + 1] - dsg.d.off[p];
                    panel_opt.steps_per_chunk = interleave_steps_panels(pnz, nsubs, (ncols + P - 1) / P > 0 ? (ncols + P - 1) / P : 1, (P + 7) / 8, f32, one, &ilv_generations);
                    if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] interleaved panels: %d panels, chunk length %d for whole generations of workgroups\n", P, panel_opt.steps_per_chunk);
                }
            }
            // rows are cut over chunks only when they are longer than half a chunk: the padding behind an interleaved chunk's non-zeros is
            // neither stored in a stream that is read nor walked (desc2.x), and a matrix without cut rows needs no fix-up launch
            if (panel_opt.split_threshold == 0) panel_opt.split_threshold = 32 * (int64_t)panel_opt.steps_per_chunk;
        }
        // interleaved panels keep their columns relative to the panel's first (the row field of the column word then has room for the
        // chunk's rows without 16-bit tags): such a part is planned and built as a matrix of the panel's width
        const int64_t pwidth = (ncols + P - 1) / P > 0 ? (ncols + P - 1) / P : 1;          // (the split's width)
        auto part_base = [&](int p) { return panel_opt.interleave > 0 ? std::min<int64_t>((int64_t)p * pwidth, std::max<int64_t>(ncols - 1, 0)) : (int64_t)0; };
        auto part_cols = [&](int p) { return panel_opt.interleave > 0 ? std::max<int64_t>(1, std::min<int64_t>(pwidth, ncols - (int64_t)p * pwidth)) : ncols; };
        if (panel_opt.interleave > 0) panel_opt.col_span = pwidth;          // (the last panel may be narrower: its column words keep the others' fields)
        std::vector<IOpt>        popts((size_t)P, panel_opt);
        std::vector<DevRows>     drs((size_t)P);
        if (dev_split) {
            // per panel: its row pointers made panel-local (a slice of the split's, minus the panel's first position), a hub
            // table of the most popular columns of its own range, and the chunk plan -- all from device arrays
            const double tp = now_s(), hub0 = in.hub_select_s;
            bool         batched = false;
            std::vector<int64_t> pcols((size_t)P);
            for (int p = 0; p < P; p++) pcols[(size_t)p] = part_cols(p);
            rc = plan_panels_batched(h, dsg.d, nsubs, pcols, f32, popts, pps, drs, &batched);      // (panels without hub tables: all plans as one submission)
            if (rc) { cvr_destroy(h); return rc; }
            if (batched && ilv_generations > 0 && panel_opt.steps_per_chunk < cvr::kIlvMaxSteps) {
                // the plan has more chunks than the estimate said and the launch would take a generation more: once more with longer chunks
                std::vector<int64_t> nch((size_t)P);
                for (int p = 0; p < P; p++) nch[(size_t)p] = pps[(size_t)p].dev_nchunks;
                const int wpb_i = panel_opt.waves_per_block > 0 ? panel_opt.waves_per_block : 4, cus_x = std::max(1, opt.cus / std::max(1, opt.xcds));
                double    fullest = 0;
                if (panel_generations(nch, (P + 7) / 8, wpb_i, cus_x, &fullest) > ilv_generations) {
                    const int S2 = (int)std::min<int64_t>(cvr::kIlvMaxSteps, ((int64_t)std::ceil(panel_opt.steps_per_chunk * fullest / ((double)ilv_generations * cus_x) * 1.02) + 3) / 4 * 4);
                    if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] interleaved panels: %.0f workgroups on the fullest XCD, more than %d generations: planned again with chunk length %d\n", fullest, ilv_generations, S2);
                    release_panel_plans(h);
                    panel_opt.steps_per_chunk = S2;
                    if (opt.split_threshold == 0) panel_opt.split_threshold = 32 * (int64_t)S2;
                    for (IOpt &o : popts) { o.steps_per_chunk = S2; o.split_threshold = panel_opt.split_threshold; }
                    pps.assign((size_t)P, PartPlan{});
                    batched = false;
                    rc = plan_panels_batched(h, dsg.d, nsubs, pcols, f32, popts, pps, drs, &batched);
                    if (rc) { cvr_destroy(h); return rc; }
                }
            }
            for (int p = 0; p < P && !batched; p++) {
                Part         &part = h->parts[(size_t)p];
                const int64_t ns = nsubs[(size_t)p], nzp = dsg.d.off[p + 1] - dsg.d.off[p];
                CREATE_TRY(hipMalloc(&part.d_rp, sizeof(int64_t) * ((size_t)ns + 1)));
                CREATE_TRY(cvr::launch_shift_rows(dsg.d.rp + dsg.d.sub0[p], ns + 1, dsg.d.off[p], part.d_rp, h->stream));
                drs[(size_t)p] = DevRows{part.d_rp, 0, nzp, h->stream, &h->plan_ws};
                rc = choose_hubs(h, part, dsg.d.ci + dsg.d.off[p], ns, ncols, f32, 0, nzp, popts[(size_t)p], pps[(size_t)p], false);
                if (rc) { cvr_destroy(h); return rc; }
                CREATE_TRY(plan_part(pps[(size_t)p], ns, part_cols(p), f32, nullptr, popts[(size_t)p], &drs[(size_t)p]));
            }
            if (batched && cvr::debug_env("fused_trace")) {
                fprintf(stderr, "[cvr] chunks per panel:");
                for (int p = 0; p < P; p++) fprintf(stderr, " %lld", (long long)pps[(size_t)p].dev_nchunks);
                fprintf(stderr, "\n");
            }
            in.plan_s += now_s() - tp - (in.hub_select_s - hub0);      // (hub selection is reported on its own)
            clk.lap("  hub tables, plans (device)");
        } else {
            // the panels' images are planned side by side (the host planner is a sequential walk per image), then built one by one
            const double tp = now_s();
            int T = (int)std::thread::hardware_concurrency();
            T = std::max(1, std::min(T, P));
            auto work = [&](int t) { for (int p = t; p < P; p += T) { pps[(size_t)p].plan_threads = 1; (void)plan_part(pps[(size_t)p], nsubs[(size_t)p], part_cols(p), f32, sp.rp[(size_t)p].data(), popts[(size_t)p]); } };
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            in.plan_s += now_s() - tp;
            clk.lap("  panels planned (parallel)");
        }
        int64_t zoff = 0, nsub = 0;
        for (int p = 0; p < P; p++) {
            Part &part = h->parts[(size_t)p];
            if (dev_split) {       // the panel's column indices and values stay where the split put them (the handle owns the split's arrays from here on)
                if (p == 0) { h->split_ci = dsg.d.ci; h->split_va = dsg.d.va; dsg.d.ci = nullptr; dsg.d.va = nullptr; }
                part.d_ci = h->split_ci + dsg.d.off[p];
                part.d_va = static_cast<uint8_t *>(h->split_va) + (size_t)dsg.d.off[p] * vsz;
                part.csr_borrowed = true;
                rc = build_part(h, part, nsubs[(size_t)p], part_cols(p), nullptr, part.d_ci, part.d_va, hipMemcpyDeviceToDevice, f32, popts[(size_t)p], &in.plan_s, &pps[(size_t)p], &drs[(size_t)p]);
            }
            else
                rc = build_part(h, part, nsubs[(size_t)p], part_cols(p), sp.rp[(size_t)p].data(), sp.ci[(size_t)p].data(),
                                sp.va[(size_t)p].data(), hipMemcpyHostToDevice, f32, popts[(size_t)p], &in.plan_s, &pps[(size_t)p]);
            if (rc) { cvr_destroy(h); return rc; }
            part.img.col_base = (uint32_t)part_base(p);
            part.zoff = zoff;
            zoff += part.yext;
            nsub += part.nrows;
        }
        if (zoff >= (int64_t)0xffffffffu) { cvr_destroy(h); return fail(CVR_ERR_INVALID, "partial-sum buffer too large for 32-bit indices"); }
        clk.lap("parts: plan, alloc, copies");
        // combine tables: the rows of every panel's sub-rows (concatenated) and, per panel, where each block of
        // kCombineRows rows starts among them
        const double   t1 = now_s();
        const uint32_t nblocks = (uint32_t)((nrows + cvr::kCombineRows - 1) / cvr::kCombineRows);
        const size_t   nboff = (size_t)P * (nblocks + 1);
        CREATE_TRY(hipEventCreateWithFlags(&h->z_free, hipEventDisableTiming));
        CREATE_TRY(hipMalloc(&h->d_z, vsz * (size_t)std::max<int64_t>(zoff, 1)));
        CREATE_TRY(hipMalloc(&h->d_block_off, sizeof(uint32_t) * nboff));
        CREATE_TRY(hipMalloc(&h->d_cpanels, sizeof(cvr::CombinePanel) * (size_t)P));
        CREATE_TRY(hipMemsetAsync(h->d_z, 0, vsz * (size_t)std::max<int64_t>(zoff, 1), h->stream));
        std::vector<cvr::CombinePanel> cps((size_t)P);
        std::vector<uint32_t>          block_off;
        CREATE_TRY(hipMalloc(&h->d_rows16, sizeof(uint16_t) * (size_t)std::max<int64_t>(nsub, 1)));
        if (dev_split) {
            h->d_rows = dsg.d.rows;        // the split's row numbers are the combine pass's, as they stand
            dsg.d.rows = nullptr;
            int64_t roff = 0;
            for (int p = 0; p < P; p++) {
                CREATE_TRY(cvr::launch_block_off(h->d_rows + roff, (uint32_t)nsubs[(size_t)p], nblocks, h->d_block_off + (size_t)p * (nblocks + 1), h->stream));
                cps[(size_t)p] = cvr::CombinePanel{static_cast<uint8_t *>(h->d_z) + (size_t)h->parts[(size_t)p].zoff * vsz, h->d_rows16 + roff};
                roff += nsubs[(size_t)p];
            }
            CREATE_TRY(cvr::launch_narrow_rows(h->d_rows, (size_t)nsub, h->d_rows16, h->stream));
        } else {
            block_off.resize(nboff);
            for (int p = 0; p < P; p++) {
                const Raw<uint32_t> &rows = sp.rows[(size_t)p];
                uint32_t            *bo = block_off.data() + (size_t)p * (nblocks + 1);
                size_t               u = 0;
                for (uint32_t b2 = 0; b2 <= nblocks; b2++) {
                    const uint64_t lim = (uint64_t)b2 * cvr::kCombineRows;
                    while (u < rows.size() && rows[u] < lim) u++;
                    bo[b2] = (uint32_t)u;
                }
            }
            CREATE_TRY(hipMalloc(&h->d_rows, sizeof(uint32_t) * (size_t)std::max<int64_t>(nsub, 1)));
            int64_t roff = 0;
            for (int p = 0; p < P; p++) {
                const Raw<uint32_t> &rows = sp.rows[(size_t)p];
                if (rows.size()) CREATE_TRY(hipMemcpyAsync(h->d_rows + roff, rows.data(), sizeof(uint32_t) * rows.size(), hipMemcpyHostToDevice, h->stream));
                cps[(size_t)p] = cvr::CombinePanel{static_cast<uint8_t *>(h->d_z) + (size_t)h->parts[(size_t)p].zoff * vsz, h->d_rows16 + roff};
                roff += (int64_t)rows.size();
            }
            CREATE_TRY(cvr::launch_narrow_rows(h->d_rows, (size_t)nsub, h->d_rows16, h->stream));
            CREATE_TRY(hipMemcpyAsync(h->d_block_off, block_off.data(), sizeof(uint32_t) * block_off.size(), hipMemcpyHostToDevice, h->stream));
        }
        in.plan_s += now_s() - t1;
        CREATE_TRY(hipMemcpyAsync(h->d_cpanels, cps.data(), sizeof(cvr::CombinePanel) * (size_t)P, hipMemcpyHostToDevice, h->stream));
        clk.lap("  combine tables enqueued");
        std::vector<cvr::FixPart> fp((size_t)P);
        for (int p = 0; p < P; p++) {
            const Part &part = h->parts[(size_t)p];
            fp[(size_t)p] = cvr::FixPart{part.img.shared, static_cast<uint8_t *>(h->d_z) + (size_t)part.zoff * vsz, (uint32_t)part.nshared, (uint32_t)part.nrows};
            h->max_nshared = std::max(h->max_nshared, (uint32_t)part.nshared);
        }
        CREATE_TRY(hipMalloc(&h->d_fixparts, sizeof(cvr::FixPart) * (size_t)P));
        CREATE_TRY(hipMemcpyAsync(h->d_fixparts, fp.data(), sizeof(cvr::FixPart) * (size_t)P, hipMemcpyHostToDevice, h->stream));
        if (setup_combine_bits(h, nsub) != CVR_OK) { cvr_destroy(h); return CVR_ERR_HIP; }
        CREATE_TRY(hipStreamSynchronize(h->stream));
        clk.lap("  tables synchronised");
        in.yext_elems = nrows + 1;
        in.image_bytes += (int64_t)(sizeof(uint32_t) * ((size_t)nsub + nboff));
    }
    clk.lap("combine tables / single part");
    // value dictionary (value_dict: <0 auto, 0 off): one code byte per slot instead of the value when the matrix has at
    // most 256 distinct values -- 12 -> 5 bytes per slot for fp64 (profiles/r01_value_dictionary.log).  The distinct
    // values are collected on the device from the uploaded CSR (a 40-MB scan takes microseconds there, milliseconds
    // on the host).
    const double t_dict0 = now_s();
    if (opt.value_dict != 0 && in.nnz > 0 && !h->preconverted) {
        if (!h->dict_scanned) {          // (single images with the automatic layout scanned together with the layout probe)
            h->dict_tab.assign(1024, ~0ull);
            for (size_t i = 0; i < h->parts.size(); i++) {
                const Part &p = h->parts[i];
                CREATE_TRY(enqueue_dict_scan(h, p.d_va, p.nnz_span - p.nnz, p.nnz_span, f32, i == 0, h->dict_tab.data(), h->dict_flags, i + 1 == h->parts.size(), h->stream));
            }
            CREATE_TRY(hipStreamSynchronize(h->stream));
            h->small_clean = false;
        }
        const std::vector<unsigned long long> &tab = h->dict_tab;
        const uint32_t                        *flags = h->dict_flags;
        if (!(flags[0] & 1u)) {
            std::vector<unsigned long long> d;
            d.push_back(0);                                                  // +0.0: the value of every pad slot
            for (unsigned long long b : tab) if (b != ~0ull) d.push_back(b);
            if (flags[0] & 2u) d.push_back(f32 ? 0xffffffffull : ~0ull);     // the all-ones pattern occurs as a value
            std::sort(d.begin(), d.end());
            d.erase(std::unique(d.begin(), d.end()), d.end());
            if (d.size() <= (size_t)cvr::kDictMax) {
                h->ndict = (uint32_t)d.size();
                std::vector<uint32_t> d32(d.begin(), d.end());
                CREATE_TRY(hipMalloc(&h->d_dict, vsz * (size_t)cvr::kDictMax));
                CREATE_TRY(hipMemsetAsync(h->d_dict, 0, vsz * (size_t)cvr::kDictMax, h->stream));
                CREATE_TRY(hipMemcpyAsync(h->d_dict, f32 ? (const void *)d32.data() : (const void *)d.data(), vsz * h->ndict, hipMemcpyHostToDevice, h->stream));
                CREATE_TRY(hipStreamSynchronize(h->stream));
            }
        }
    }
    in.value_dict = (int32_t)h->ndict;
    if (!h->preconverted) in.dict_s = now_s() - t_dict0;
    clk.lap("value dictionary scan");
    if (!h->preconverted) for (Part &p : h->parts) { rc = finish_part(h, p); if (rc) { cvr_destroy(h); return rc; } }
    for (const Part &p : h->parts) in.hub_entries = std::max<int32_t>(in.hub_entries, (int32_t)p.img.hub_n);
    // column panels whose images are plain (one chunk per workgroup, no LDS tables): eight panels per launch, each on the XCD of
    // its workgroups, so that an L2 holds one slice of x at a time and every line of x is fetched by one XCD only
    if (h->paneled() && xcd_panels) {
        bool plain = true;
        for (const Part &p : h->parts) plain = plain && (p.img.ilv ? p.img.wpb == h->parts[0].img.wpb : p.img.wpb <= 1) && p.img.ilv == h->parts[0].img.ilv && p.img.hub_n == 0 && p.img.win_elems == 0 && p.img.phases == h->parts[0].img.phases && p.img.tag16 == h->parts[0].img.tag16 && !p.img.c16 && p.img.S == h->parts[0].img.S;
        if (plain) {
            const size_t per_round = cvr::debug_env("xcd_panels_debug") ? (size_t)atoi(cvr::debug_env("xcd_panels_debug")) : 8;      // (diagnostics: fewer panels side by side)
            const size_t rounds = (h->parts.size() + per_round - 1) / per_round;
            std::vector<cvr::PanelArgs> pa(rounds * 8, cvr::PanelArgs{nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u, nullptr, 0u, 0u});
            h->multi_chunks.assign(rounds, 0u);
            // which panel runs where: the heaviest first, each to the XCD with the least work so far that still has a round free (the XCDs
            // go through their panels independently: what counts is every XCD's sum, not the rounds'); equal-width panels of a real graph
            // differ in non-zeros
            std::vector<size_t> slot_of(h->parts.size());
            if (per_round == 8) {
                std::vector<size_t> order(h->parts.size());
                for (size_t j = 0; j < order.size(); j++) order[j] = j;
                std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return h->parts[a].img.nchunks > h->parts[b].img.nchunks; });
                uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                size_t   used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t j : order) {
                    int best = -1;
                    for (int x = 0; x < 8; x++)
                        if (used[x] < rounds && (best < 0 || load[x] < load[best])) best = x;
                    slot_of[j] = used[best] * 8 + (size_t)best;
                    used[best]++;
                    load[best] += h->parts[j].img.nchunks;
                }
            } else
                for (size_t j = 0; j < slot_of.size(); j++) slot_of[j] = (j / per_round) * 8 + j % per_round;
            for (size_t j = 0; j < h->parts.size(); j++) {
                const Part &p = h->parts[j];
                const size_t i = slot_of[j];
                h->parts[j].multi_slot = (int32_t)i;
                pa[i] = cvr::PanelArgs{p.img.stream, p.img.desc, p.img.target, static_cast<uint8_t *>(h->d_z) + (size_t)p.zoff * vsz, p.img.nchunks, p.img.ystage, p.img.desc2, p.img.col_base, p.img.pad_col, p.img.gbase, 0u, 0u};
                h->multi_chunks[i / 8] = std::max(h->multi_chunks[i / 8], p.img.nchunks);
                h->multi_ystage = std::max(h->multi_ystage, p.img.ystage);
                if (cvr::debug_env("xcd_panels_trace")) fprintf(stderr, "[xcd panels] part %zu slot %zu nchunks %u ystage %u S %d G %d zoff %lld yext %lld nshared %u stream %p\n", j, i, p.img.nchunks, p.img.ystage, p.img.S, p.img.G, (long long)p.zoff, (long long)p.yext, p.img.nshared, (void *)p.img.stream);
            }
            CREATE_TRY(hipMalloc(&h->d_multi, sizeof(cvr::PanelArgs) * pa.size()));
            CREATE_TRY(hipMemcpy(h->d_multi, pa.data(), sizeof(cvr::PanelArgs) * pa.size(), hipMemcpyHostToDevice));
        }
    }
    if (!h->z_free && in.hub_entries) CREATE_TRY(hipEventCreateWithFlags(&h->z_free, hipEventDisableTiming));
    { rc = setup_fuse(h); if (rc) { cvr_destroy(h); return rc; } }
    in.steps_per_chunk = h->parts[0].img.S;
    in.col_phases = h->parts[0].img.ilv ? 1 : (int32_t)h->parts[0].img.phases;      // (an interleaved image is planned like one with phases, but has none)
    in.waves_per_block = (int32_t)h->parts[0].img.wpb; in.x_window = (int32_t)h->parts[0].img.win_elems;
    in.lds_bytes = (int32_t)cvr::spmv_lds_bytes(h->parts[0].img);
    in.narrow_cols = h->parts[0].img.c16 ? 1 : 0;
    in.hub_reorder = h->parts[0].img.order_n ? 1 : 0;
    in.row_tags16 = h->parts[0].img.tag16 ? 1 : 0;
    in.row_bands = 1;
    in.piece_max = (int32_t)h->parts[0].img.piece_max;
    in.interleave = h->parts[0].img.ilv ? 1 : 0;
    in.gang = (int32_t)h->parts[0].img.gang;
    in.chunk_row_cap = h->parts[0].img.phases > 1 ? (int64_t)h->parts[0].img.ystage - 1 : 0;
    for (const Part &p : h->parts) {
        in.nchunks += p.nchunks; in.nshared += p.nshared; in.nslots += p.nchunks * 64 * p.img.S;
        in.image_bytes += (int64_t)(p.stream_bytes + (size_t)p.nchunks * (16 + 64) + (size_t)p.nshared * 24);
    }
    {   // room for the conversion-time segment table of images with column phases (cvr_preprocess takes it over and releases it)
        size_t n1 = 0, nch = 0;
        for (const Part &p : h->parts)
            if (p.img.phases > 1 && !p.img.ilv && p.nchunks > 0) { n1 = std::max(n1, (size_t)p.nchunks * (size_t)cvr::kLanes * (size_t)p.img.S); nch = std::max(nch, (size_t)p.nchunks); }
        if (n1 > 0 && n1 < ((size_t)1 << 32) && !h->preconverted) {
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t bytes = up(sizeof(int64_t) * n1) + up(sizeof(uint32_t) * n1) + up(sizeof(uint16_t) * n1) + up(sizeof(uint32_t) * (nch + 1)) + 256;
            if (hipMalloc(&h->seg_arena, bytes) == hipSuccess) h->seg_arena_bytes = bytes; else { (void)hipGetLastError(); h->seg_arena = nullptr; }
        }
    }
    ilv_runtime_settings(h);
    in.spmv_launches = h->d_multi ? 1 : (int32_t)h->parts.size();
    if (!h->d_err) CREATE_TRY(hipMalloc(&h->d_err, sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(&h->d_x, vsz * (size_t)in.x_elems));
    CREATE_TRY(hipMalloc(&h->d_y, vsz * (size_t)in.yext_elems));
    CREATE_TRY(hipMemsetAsync(h->d_x, 0, vsz * (size_t)in.x_elems, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_y, 0, vsz * (size_t)in.yext_elems, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
    CREATE_TRY(hipStreamSynchronize(h->stream));   // the caller may free its CSR when this returns
    in.upload_s = now_s() - t_up0 - in.plan_s - in.probe_s - panel_rule_s - in.hub_select_s;      // (hub selection and the layout probe are reported on their own)
    in.plan_s += panel_rule_s;
    cvr::free_plan_scratch(h->plan_ws);
    (void)hipFree(h->d_small); h->d_small = nullptr;
    std::vector<unsigned long long>().swap(h->dict_tab);
    clk.lap("images, x, y");
#undef CREATE_TRY
    *out = h;
    return CVR_OK;
}

int cvr_preprocess(cvr_handle *h, int keep_csr, double *seconds)
{
    cvr::debug_refresh();
    if (!h) return fail(CVR_ERR_INVALID, "handle is null");
    if (h->parts.empty() || !h->parts[0].d_rp) return fail(CVR_ERR_STATE, "the device CSR was already released: cvr_preprocess runs once unless keep_csr was set");
    Range range("cvr_preprocess (CSR -> CVR64)");
    const double t_wall0 = now_s();
    HIP_TRY(hipSetDevice(h->device));
    if (h->preconverted) {          // cvr_create converted behind the planner, in one submission (cvr_fused.hip): its times are in cvr_info already
        h->preconverted = false;
        if (seconds) *seconds = h->info.convert_s;
        h->info.preprocess_wall_s = now_s() - t_wall0;
        h->converted = true;
        if (!keep_csr) { for (Part &p : h->parts) p.release_csr(); h->release_split(); }
        return CVR_OK;
    }
    if (h->events.size() < 2) {
        h->events.resize(2);
        HIP_TRY(hipEventCreate(&h->events[0]));
        HIP_TRY(hipEventCreate(&h->events[1]));
    }
    const hipEvent_t e0 = h->events[0], e1 = h->events[1];
    HIP_TRY(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
    struct SegGuard { cvr::SegTable t; void *arena = nullptr; uint8_t *codes = nullptr; ~SegGuard() { (void)hipFree(arena); (void)hipFree(codes); } } sg;      // (one allocation: six cost six times the call)
    Part &p0 = h->parts[0];
    // column phases: the segment table (conversion-time only) gives every chunk room for as many segments as it has slots, so
    // that counting and filling are one kernel per chunk and the conversion follows without the host in between; the images of a
    // handle (column panels) are converted one after the other and share the table, sized for the largest
    size_t seg_n1 = 0, seg_chunks = 0;
    int64_t ilv_nnz = 0;             // interleaved images: converted together behind the loop (one sort over all their non-zeros)
    bool     any_ilv = false, any_gang = false;
    uint32_t ilv_chunks = 0;
    for (const Part &p : h->parts)
        if (p.img.ilv) { ilv_nnz += p.nnz; ilv_chunks += (uint32_t)p.nchunks; any_ilv = true; any_gang = any_gang || p.img.gang != 0; }
        else if (p.img.phases > 1 && p.nchunks > 0) {
            seg_n1 = std::max(seg_n1, (size_t)p.nchunks * (size_t)cvr::kLanes * (size_t)p.img.S);
            seg_chunks = std::max(seg_chunks, (size_t)p.nchunks);
        }
    struct IlvGuard { void *p = nullptr; ~IlvGuard() { (void)hipFree(p); } } ilv;
    const size_t ilv_scratch = any_ilv ? cvr::convert_interleaved_scratch(ilv_nnz, ilv_chunks, any_gang) : 0;
    if (ilv_scratch) HIP_TRY(hipMalloc(&ilv.p, ilv_scratch));
    std::vector<const cvr::DeviceImage *> ilv_imgs;
    std::vector<cvr::DeviceCsr>           ilv_csrs;
    std::vector<int64_t>                  ilv_n0, ilv_n1;
    const bool phased = seg_n1 > 0;
    if (phased) {
        cvr::SegTable &t = sg.t;
        const size_t n1 = seg_n1;
        if (n1 >= ((size_t)1 << 32)) return fail(CVR_ERR_INVALID, "col_phases: more than 2^32 slots in one image");
        auto         up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_begin = 0, o_len = o_begin + up(sizeof(int64_t) * n1), o_row = o_len + up(sizeof(uint32_t) * n1), o_cnt = o_row + up(sizeof(uint16_t) * n1),
                     o_flags = o_cnt + up(sizeof(uint32_t) * (seg_chunks + 1));
        if (h->seg_arena && h->seg_arena_bytes >= o_flags + 256) { sg.arena = h->seg_arena; h->seg_arena = nullptr; h->seg_arena_bytes = 0; }      // (cvr_create made room)
        else HIP_TRY(hipMalloc(&sg.arena, o_flags + 256));
        uint8_t *a = static_cast<uint8_t *>(sg.arena);
        t.begin = reinterpret_cast<int64_t *>(a + o_begin); t.len = reinterpret_cast<uint32_t *>(a + o_len); t.row = reinterpret_cast<uint16_t *>(a + o_row);
        t.cnt = reinterpret_cast<uint32_t *>(a + o_cnt); t.flags = reinterpret_cast<uint32_t *>(a + o_flags);
        HIP_TRY(hipMemsetAsync(t.flags, 0, sizeof(uint32_t) * 2, h->stream));
    }
    // the window choice needs nothing of the conversion (and nothing is pending on the handle's stream: cvr_create ended with a
    // synchronisation): a single image's runs beside it on the side stream
    hipStream_t wstream = h->paneled() || !p0.img.win_elems ? h->stream : side_stream(h->device);
    if (!wstream) wstream = h->stream;
    HIP_TRY(hipEventRecord(e0, h->stream));
    uint32_t              seg_flags[2] = {0, 0};
    std::vector<uint32_t> seg_totals(h->parts.size(), 0u);
    for (Part &p : h->parts) {
        cvr::DeviceCsr csr;
        csr.row_ptr = p.d_rp; csr.col_idx = p.d_ci; csr.vals = p.d_va; csr.nz_begin = p.d_nzb; csr.pad_cnt = p.d_pad;
        if (wstream != h->stream) HIP_TRY(cvr::launch_window(p.img, csr, wstream));
        if (p.img.ilv) {
            if (p.nchunks > 0) { ilv_imgs.push_back(&p.img); ilv_csrs.push_back(csr); ilv_n0.push_back(p.nnz_span - p.nnz); ilv_n1.push_back(p.nnz_span); }
        } else if (p.img.phases > 1 && p.nchunks > 0) {
            cvr::SegTable &t = sg.t;
            if (h->d_dict && cvr::seg_table_packed_ok(p.img) && !cvr::debug_env("no_dict_codes")) {      // the values as dictionary codes first: the converter then reads a byte per value (cvr_convert.hip: convert_lds_kernel)
                (void)hipFree(sg.codes); sg.codes = nullptr;
                HIP_TRY(hipMalloc(&sg.codes, std::max<size_t>((size_t)p.nnz_span, 1)));
                HIP_TRY(cvr::launch_dict_codes(p.d_va, p.nnz_span - p.nnz, p.nnz_span, p.img.f32, h->d_dict, h->ndict, sg.codes, h->d_err, h->stream));
                csr.codes = sg.codes;
            }
            HIP_TRY(cvr::launch_seg_build(p.img, csr, t, h->stream));
            HIP_TRY(cvr::launch_convert(p.img, csr, h->d_err, h->stream, &t));
            HIP_TRY(hipMemcpyAsync(seg_flags, t.flags, sizeof(seg_flags), hipMemcpyDeviceToHost, h->stream));      // (the flags of all images so far)
            HIP_TRY(hipMemcpyAsync(&seg_totals[(size_t)(&p - h->parts.data())], t.cnt + p.nchunks, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        } else {
            HIP_TRY(cvr::launch_convert(p.img, csr, h->d_err, h->stream));
        }
        if (wstream == h->stream) HIP_TRY(cvr::launch_window(p.img, csr, h->stream));
    }
    auto convert_ilv = [&]() -> int {
        // groups of up to 64 images with equal layout parameters (the panels of one matrix: one group)
        for (size_t i0 = 0; i0 < ilv_imgs.size();) {
            size_t i1 = i0 + 1;
            while (i1 < ilv_imgs.size() && i1 - i0 < 64 && ilv_imgs[i1]->G == ilv_imgs[i0]->G && ilv_imgs[i1]->tag16 == ilv_imgs[i0]->tag16 && ilv_imgs[i1]->col_bits == ilv_imgs[i0]->col_bits && ilv_imgs[i1]->gang == ilv_imgs[i0]->gang) i1++;
            HIP_TRY(cvr::launch_convert_interleaved(ilv_imgs.data() + i0, ilv_csrs.data() + i0, ilv_n0.data() + i0, ilv_n1.data() + i0, (int)(i1 - i0), h->d_err, ilv.p, ilv_scratch, h->stream));
            i0 = i1;
        }
        return CVR_OK;
    };
    if (!ilv_imgs.empty()) { const int rci = convert_ilv(); if (rci) return rci; }
    HIP_TRY(hipEventRecord(e1, h->stream));
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, h->d_err, sizeof(err), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if ((err & 8u) && any_gang) {
        // gang chunks: a column lies further than 2^17 from its group's first (a panel with few non-zeros, or one with wide empty column ranges): the
        // images take 16-bit tags instead -- their streams are allocated again (a group is 512 bytes longer) and converted once more
        if (cvr::debug_env("fused_trace")) fprintf(stderr, "[cvr] gang chunks: a column offset beyond %d bits: converting again with 16-bit tags\n", cvr::kGangOffBits);
        for (Part &p : h->parts) {
            if (!p.img.gang || p.img.tag16) continue;
            (void)hipFree(p.img.stream); p.img.stream = nullptr;
            if (p.img.gbase) { (void)hipFree(p.img.gbase); p.img.gbase = nullptr; }
            p.img.tag16 = true; p.img.col_bits = 31;
            p.img.col_mask = h->opt_used.debug_col_mask ? (cvr::kColMask & (uint32_t)h->opt_used.debug_col_mask) : cvr::kColMask;
            const size_t before = p.stream_bytes;
            const int    rcf = finish_part(h, p);
            if (rcf) return rcf;
            h->info.image_bytes += (int64_t)p.stream_bytes - (int64_t)before;
        }
        if (h->d_multi && !h->multi_chunks.empty()) {          // the panels' launch table names the streams
            std::vector<cvr::PanelArgs> pa(h->multi_chunks.size() * 8);
            HIP_TRY(hipMemcpy(pa.data(), h->d_multi, sizeof(cvr::PanelArgs) * pa.size(), hipMemcpyDeviceToHost));
            for (const Part &p : h->parts) if (p.multi_slot >= 0 && (size_t)p.multi_slot < pa.size()) { pa[(size_t)p.multi_slot].stream = p.img.stream; pa[(size_t)p.multi_slot].gbase = p.img.gbase; }
            HIP_TRY(hipMemcpy(h->d_multi, pa.data(), sizeof(cvr::PanelArgs) * pa.size(), hipMemcpyHostToDevice));
        }
        h->info.row_tags16 = 1;
        HIP_TRY(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
        HIP_TRY(hipEventRecord(e0, h->stream));
        { const int rci = convert_ilv(); if (rci) return rci; }
        HIP_TRY(hipEventRecord(e1, h->stream));
        HIP_TRY(hipMemcpyAsync(&err, h->d_err, sizeof(err), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        ilv_runtime_settings(h);
    }
    if (wstream != h->stream) HIP_TRY(hipStreamSynchronize(wstream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    h->info.convert_s = ms * 1e-3;
    h->info.lds_bytes = (int32_t)cvr::spmv_lds_bytes(h->parts[0].img);      // column phases: the segment-row copy is sized by now
    if (seconds) *seconds = ms * 1e-3;
    if (seg_flags[0] & 1u) return fail(CVR_ERR_INVALID, "col_phases needs the column indices of every row in ascending order");
    if (seg_flags[0] & 2u) return fail(CVR_ERR_INTERNAL, "segment table: a chunk of the plan exceeds the launch's chunk length");
    if (err) return fail(CVR_ERR_INTERNAL, "device converter self-check failed (flags 0x%x)", err);
    h->info.nsegments = 0;
    for (uint32_t v : seg_totals) h->info.nsegments += v;
    h->info.preprocess_wall_s = now_s() - t_wall0;
    h->converted = true;
    if (!keep_csr) { for (Part &p : h->parts) p.release_csr(); h->release_split(); }
    return CVR_OK;
}

int cvr_get_info(const cvr_handle *h, cvr_info *info)
{
    if (!h || !info) return fail(CVR_ERR_INVALID, "null argument");
    *info = h->info;
    return CVR_OK;
}

int cvr_destroy(cvr_handle *h)
{
    if (!h) return CVR_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (Part &p : h->parts) p.release_all();
    h->release_split();
    if (h->panel_tables) (void)hipFree(h->panel_tables);
    if (h->fuse_mem) (void)hipFree(h->fuse_mem);
    cvr::free_plan_scratch(h->plan_ws);
    if (h->seg_arena) (void)hipFree(h->seg_arena);
    if (h->d_small) (void)hipFree(h->d_small);
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    if (h->z_free) (void)hipEventDestroy(h->z_free);
    for (void *p : {(void *)h->d_err, h->d_z, (void *)h->d_rows, (void *)h->d_rows16, (void *)h->d_cbits, (void *)h->d_cut, (void *)h->d_block_off, (void *)h->d_cpanels, (void *)h->d_fixparts, (void *)h->d_multi, h->d_dict, h->d_x, h->d_y}) if (p) (void)hipFree(p);
    release_stream(h->device, h->stream);
    delete h;
    return CVR_OK;
}

void *cvr_x_device(cvr_handle *h) { return h ? h->d_x : nullptr; }
void *cvr_y_device(cvr_handle *h) { return h ? h->d_y : nullptr; }
void *cvr_stream(cvr_handle *h) { return h ? (void *)h->stream : nullptr; }

int cvr_spmv_device(cvr_handle *h, const void *x_dev, void *y_dev, void *stream)
{
    if (!h || !x_dev || !y_dev) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    HIP_TRY(hipSetDevice(h->device));          // the NULL stream means the current device's
    HIP_TRY(run_spmv(h, x_dev, y_dev, (hipStream_t)stream));
    return CVR_OK;
}

int cvr_spmv_device_repeat(cvr_handle *h, const void *x_dev, void *y_dev, void *stream, int n)
{
    if (!h || !x_dev || !y_dev) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    const hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < n; i++) HIP_TRY(run_spmv(h, x_dev, y_dev, st));
    return CVR_OK;
}

int cvr_spmv_bench(cvr_handle *h, int warmup, int iters, double *mean_s)
{
    if (!h || iters < 1) return fail(CVR_ERR_INVALID, "bad argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    HIP_TRY(hipSetDevice(h->device));
    if (h->events.size() < 2) {
        h->events.resize(2);
        HIP_TRY(hipEventCreate(&h->events[0]));
        HIP_TRY(hipEventCreate(&h->events[1]));
    }
    for (int i = 0; i < warmup; i++) HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));
    HIP_TRY(hipEventRecord(h->events[0], h->stream));
    for (int i = 0; i < iters; i++) HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));
    HIP_TRY(hipEventRecord(h->events[1], h->stream));
    HIP_TRY(hipEventSynchronize(h->events[1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->events[0], h->events[1]));
    if (mean_s) *mean_s = ms * 1e-3 / iters;
    return CVR_OK;
}

int cvr_spmv(cvr_handle *h, const void *x_host, void *y_host, int iters, cvr_timing *tm)
{
    if (!h || !x_host || !y_host) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_spmv before cvr_preprocess");
    if (iters < 1) iters = 1;
    Range range("cvr_spmv (h2d x, timed launches, d2h y)");
    HIP_TRY(hipSetDevice(h->device));
    const size_t need = (size_t)iters + 1;
    while (h->events.size() < need) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->events.push_back(e);
    }
    double t0 = now_s();
    if (h->info.ncols) HIP_TRY(hipMemcpyAsync(h->d_x, x_host, h->vsz * (size_t)h->info.ncols, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double h2d = now_s() - t0;
    HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));   // warm-up, untimed
    HIP_TRY(hipEventRecord(h->events[0], h->stream));
    for (int i = 0; i < iters; i++) {
        HIP_TRY(run_spmv(h, h->d_x, h->d_y, h->stream));
        HIP_TRY(hipEventRecord(h->events[(size_t)i + 1], h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    t0 = now_s();
    if (h->info.nrows) HIP_TRY(hipMemcpyAsync(y_host, h->d_y, h->vsz * (size_t)h->info.nrows, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const double d2h = now_s() - t0;
    if (tm) {
        memset(tm, 0, sizeof(*tm));
        tm->iters = iters; tm->h2d_s = h2d; tm->d2h_s = d2h;
        double              sum = 0;
        std::vector<double> ts((size_t)iters);
        for (int i = 0; i < iters; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, h->events[(size_t)i], h->events[(size_t)i + 1]));
            ts[(size_t)i] = ms * 1e-3;
            sum += ts[(size_t)i];
        }
        float tot = 0;
        HIP_TRY(hipEventElapsedTime(&tot, h->events[0], h->events[(size_t)iters]));
        std::sort(ts.begin(), ts.end());
        tm->mean_s = sum / iters; tm->min_s = ts.front(); tm->max_s = ts.back(); tm->median_s = ts[ts.size() / 2]; tm->total_s = tot * 1e-3;
        // one GPU: no exchange step (include/cvr_amd.h, cvr_timing)
        tm->step_mean_s = tm->mean_s; tm->step_min_s = tm->min_s; tm->step_median_s = tm->median_s; tm->step_max_s = tm->max_s; tm->gather_mean_s = 0;
    }
    return CVR_OK;
}

int cvr_device_copy_bench(int device, int64_t bytes, int iters, double *gbs)
{
    if (bytes < 16 || iters < 1) return fail(CVR_ERR_INVALID, "bad argument");
    if (device < 0 || device >= cvr_device_count()) return fail(CVR_ERR_NO_DEVICE, "device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    void       *a = nullptr, *b = nullptr;
    hipStream_t st;
    hipEvent_t  e0, e1;
    HIP_TRY(hipMalloc(&a, (size_t)bytes));
    HIP_TRY(hipMalloc(&b, (size_t)bytes));
    HIP_TRY(hipStreamCreate(&st));
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipMemsetAsync(a, 1, (size_t)bytes, st));
    for (int i = 0; i < 3; i++) HIP_TRY(cvr::launch_copy(a, b, (size_t)bytes, st));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) HIP_TRY(cvr::launch_copy(a, b, (size_t)bytes, st));
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (gbs) *gbs = 2.0 * (double)(bytes / 16 * 16) * iters / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st); (void)hipFree(a); (void)hipFree(b);
    return CVR_OK;
}

int cvr_export_image(cvr_handle *h, void *stream_image, uint32_t *desc, uint8_t *target, int64_t *shared)
{
    if (!h) return fail(CVR_ERR_INVALID, "handle is null");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_export_image before cvr_preprocess");
    if (h->paneled()) return fail(CVR_ERR_STATE, "cvr_export_image exports one image: create the handle with col_panels = 1");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const Part  &p = h->parts[0];
    const size_t nc = (size_t)p.nchunks;
    if (stream_image && p.stream_bytes) HIP_TRY(hipMemcpy(stream_image, p.img.stream, p.stream_bytes, hipMemcpyDeviceToHost));
    if (desc && nc) HIP_TRY(hipMemcpy(desc, p.img.desc, 16 * nc, hipMemcpyDeviceToHost));
    if (target && nc) HIP_TRY(hipMemcpy(target, p.img.target, 64 * nc, hipMemcpyDeviceToHost));
    if (shared && p.nshared) HIP_TRY(hipMemcpy(shared, p.img.shared, 24 * (size_t)p.nshared, hipMemcpyDeviceToHost));
    return CVR_OK;
}

int cvr_export_gang(cvr_handle *h, uint32_t *group_first_cols, uint32_t *desc2)
{
    if (!h) return fail(CVR_ERR_INVALID, "handle is null");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_export_gang before cvr_preprocess");
    if (h->paneled()) return fail(CVR_ERR_STATE, "cvr_export_gang exports one image: create the handle with col_panels = 1");
    const Part  &p = h->parts[0];
    if (!p.img.gang) return fail(CVR_ERR_STATE, "cvr_export_gang: the image has no gang chunks");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t nc = (size_t)p.nchunks, ng = nc * (size_t)p.img.G;
    if (group_first_cols && ng) {
        if (p.img.gbase) HIP_TRY(hipMemcpy(group_first_cols, p.img.gbase, sizeof(uint32_t) * ng, hipMemcpyDeviceToHost));
        else memset(group_first_cols, 0, sizeof(uint32_t) * ng);
    }
    if (desc2 && nc) HIP_TRY(hipMemcpy(desc2, p.img.desc2, 8 * nc, hipMemcpyDeviceToHost));
    return CVR_OK;
}

}  // extern "C"

// cvr_image_io.hip -- the converted CVR64 image of a handle on disk (cvr_save_image / cvr_load_image): SURVEY.md 8(f) item 1, the
// ".cvrbin" half.  The image is a deterministic function of (matrix, options, device geometry, library version), so a second run
// on the same .mtx can skip the analysis passes, the planner and the converter (the reference repeats its pre_processing on every
// run: spmv.cpp:1857, timed at :1009) and load the image straight into device memory.  A file is only accepted under the key it
// was written with: the identity of the source file (cvr_source_key), the options, the device's CU / XCD counts, the format
// version and the library version; anything else is CVR_ERR_STATE and the caller converts again.
#include "cvr_internal.h"

using namespace cvrh;

namespace {

constexpr uint64_t kImgMagic = 0x3130474d49525643ull;      // "CVRIMG01"
constexpr uint32_t kImgVersion = 10;                       // bump when DeviceImage / the handle's tables change

struct ImgKey {
    uint64_t       magic;
    uint32_t       version, vsz;
    cvr_source_key source;
    int64_t        opt[20];          // the options, field by field (no padding bytes in the key)
    int32_t        cus, xcds, reserved[2];
    char           lib[48];
};

struct Writer {
    FILE *f = nullptr;
    bool  ok = true;
    void  raw(const void *p, size_t n) { if (ok && n) ok = fwrite(p, 1, n, f) == n; }
    template <typename T> void pod(const T &v) { raw(&v, sizeof(T)); }
    // a device array: byte count, then the bytes (staged through `host`)
    void dev(const void *d, size_t bytes, std::vector<uint8_t> &host)
    {
        const uint64_t n = d ? bytes : 0;
        pod(n);
        if (!ok || !n) return;
        host.resize((size_t)n);
        if (hipMemcpy(host.data(), d, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; return; }
        raw(host.data(), (size_t)n);
    }
};

struct Reader {
    FILE *f = nullptr;
    bool  ok = true;
    void  raw(void *p, size_t n) { if (ok && n) ok = fread(p, 1, n, f) == n; }
    template <typename T> void pod(T &v) { raw(&v, sizeof(T)); }
    // a device array of `bytes` (exactly what the writer stored): allocated here, filled through the pinned staging buffer
    // `want`: the byte count the image's scalars imply -- a file that stores another one is damaged (or crafted) and is refused before
    // anything is allocated; `optional`: the array may be absent (0 bytes stored)
    template <typename P> hipError_t dev(P *&d, uint8_t *pinned, size_t pinned_bytes, hipStream_t st, uint64_t want, bool optional = false, size_t min_alloc = 1)
    {
        uint64_t n = 0;
        pod(n);
        d = nullptr;
        if (!ok) return hipErrorUnknown;
        if (!(n == want || (optional && n == 0))) { ok = false; return hipErrorUnknown; }
        if (!n) return hipSuccess;
        hipError_t e = hipMalloc(&d, std::max<size_t>((size_t)n, min_alloc));
        for (uint64_t off = 0; e == hipSuccess && off < n; off += pinned_bytes) {
            const size_t c = (size_t)std::min<uint64_t>(pinned_bytes, n - off);
            e = hipStreamSynchronize(st);                       // (the staging buffer is free again)
            if (e != hipSuccess) break;
            raw(pinned, c);
            if (!ok) return hipErrorUnknown;
            e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(d) + off, pinned, c, hipMemcpyHostToDevice, st);
        }
        return e;
    }
};

void make_key(ImgKey &k, const cvr_source_key *src, const IOpt &o, size_t vsz)
{
    memset(&k, 0, sizeof(k));
    k.magic = kImgMagic; k.version = kImgVersion; k.vsz = (uint32_t)vsz;
    if (src) k.source = *src;
    const int64_t ov[] = {0 /* device: not part of the identity */, o.steps_per_chunk, o.split_threshold, o.xcd_swizzle, o.x_window, o.waves_per_block, o.col_panels, o.value_dict,
                          o.col_phases, o.hub_table, o.narrow_cols, o.hub_reorder, o.row_tags16, o.piece_max, o.debug_col_mask, o.interleave};
    static_assert(sizeof(ov) <= sizeof(k.opt), "options fit the key");
    memcpy(k.opt, ov, sizeof(ov));
    k.cus = o.cus; k.xcds = o.xcds;
    snprintf(k.lib, sizeof(k.lib), "%s", cvr_version());
}

// the scalar part of a DeviceImage (everything but its pointers), field by field so that the file does not depend on the struct's layout
struct ImgScalars {
    int32_t  S, G, f32, xcd_swizzle, c16, tag16;
    uint32_t nchunks, nrows, pad_col, nshared, ystage, ndict, wpb, win_elems, col_mask, phases, phase_width, col_bits, piece_max, hub_n, order_n, ncus;
    int64_t  part_nrows, part_nnz, part_nnz_span, part_nchunks, part_nshared, part_yext, part_zoff;
    uint64_t stream_bytes;
    int32_t  multi_slot, ilv;      // where the panel stands in the rounds of eight (-1: none)
    uint32_t col_base, gang;          // gang: wavefronts that walk one common list (0 = none)
};

}  // namespace

extern "C" {

int cvr_save_image(cvr_handle *h, const char *path, const cvr_source_key *key)
{
    if (!h || !path) return fail(CVR_ERR_INVALID, "null argument");
    if (!h->converted) return fail(CVR_ERR_STATE, "cvr_save_image before cvr_preprocess");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const std::string tmp = std::string(path) + ".tmp";
    Writer w;
    w.f = fopen(tmp.c_str(), "wb");
    if (!w.f) return fail(CVR_ERR_IO, "cannot write %s", tmp.c_str());
    std::vector<uint8_t> host;
    ImgKey k;
    make_key(k, key, h->opt_used, h->vsz);
    w.pod(k);
    w.pod(h->info);
    const uint32_t nparts = (uint32_t)h->parts.size(), nrounds = (uint32_t)h->multi_chunks.size();
    w.pod(nparts); w.pod(h->ndict); w.pod(h->max_nshared); w.pod(h->multi_ystage); w.pod(nrounds);
    const uint32_t has_multi = h->d_multi ? 1u : 0u;
    w.pod(has_multi);
    for (uint32_t r = 0; r < nrounds; r++) w.pod(h->multi_chunks[r]);
    w.dev(h->d_dict, h->d_dict ? h->vsz * (size_t)cvr::kDictMax : 0, host);
    int64_t nsub = 0;
    for (const Part &p : h->parts) {
        const cvr::DeviceImage &g = p.img;
        ImgScalars s;
        memset(&s, 0, sizeof(s));
        s.S = g.S; s.G = g.G; s.f32 = g.f32; s.xcd_swizzle = g.xcd_swizzle; s.c16 = g.c16; s.tag16 = g.tag16;
        s.nchunks = g.nchunks; s.nrows = g.nrows; s.pad_col = g.pad_col; s.nshared = g.nshared; s.ystage = g.ystage; s.ndict = g.ndict;
        s.wpb = g.wpb; s.win_elems = g.win_elems; s.col_mask = g.col_mask; s.phases = g.phases; s.phase_width = g.phase_width; s.col_bits = g.col_bits;
        s.piece_max = g.piece_max; s.hub_n = g.hub_n; s.order_n = g.order_n; s.ncus = g.ncus;
        s.part_nrows = p.nrows; s.part_nnz = p.nnz; s.part_nnz_span = p.nnz_span; s.part_nchunks = p.nchunks; s.part_nshared = p.nshared; s.part_yext = p.yext; s.part_zoff = p.zoff;
        s.stream_bytes = p.stream_bytes;
        s.multi_slot = p.multi_slot; s.ilv = g.ilv ? 1 : 0; s.col_base = g.col_base; s.gang = g.gang;
        w.pod(s);
        const size_t nc = (size_t)g.nchunks;
        const size_t slack = 8 * (size_t)cvr::group_bytes(g.f32, g.dict != nullptr, g.c16, g.tag16);
        (void)slack;
        w.dev(g.stream, p.stream_bytes, host);
        w.dev(g.desc, 16 * nc, host);
        w.dev(g.target, 64 * nc, host);
        w.dev(g.shared, 24 * (size_t)g.nshared, host);
        w.dev(g.win_base, sizeof(uint32_t) * (nc / std::max<uint32_t>(g.wpb, 1u) + 1), host);
        w.dev(g.desc2, g.desc2 ? 8 * nc : 0, host);
        w.dev(g.cbase, g.cbase ? sizeof(uint32_t) * nc : 0, host);
        w.dev(g.gbase, g.gbase ? sizeof(uint32_t) * (nc * (size_t)g.G + 4096) : 0, host);
        w.dev(g.hub_cols, g.hub_cols ? sizeof(int32_t) * (size_t)(g.order_n ? g.order_n : g.hub_n) : 0, host);
        nsub += p.nrows;
    }
    if (h->paneled()) {
        const uint32_t nblocks = (uint32_t)((h->info.nrows + cvr::kCombineRows - 1) / cvr::kCombineRows);
        w.dev(h->d_rows, sizeof(uint32_t) * (size_t)std::max<int64_t>(nsub, 1), host);
        w.dev(h->d_block_off, sizeof(uint32_t) * (size_t)nparts * (nblocks + 1), host);
    }
    const uint64_t tail = kImgMagic;      // (a truncated file does not end with it)
    w.pod(tail);
    const bool ok = fclose(w.f) == 0 && w.ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { (void)remove(tmp.c_str()); return fail(CVR_ERR_IO, "writing %s failed", path); }
    return CVR_OK;
}

int cvr_load_image(cvr_handle **out, const char *path, const cvr_source_key *expect, const cvr_options *opt_in, double *seconds)
{
    cvr::debug_refresh();
    if (!out) return fail(CVR_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!path) return fail(CVR_ERR_INVALID, "null argument");
    const double t0 = now_s();
    IOpt opt = make_iopt(opt_in);
    const int ndev = cvr_device_count();
    if (ndev <= 0) return fail(CVR_ERR_NO_DEVICE, "no HIP device visible: libcvr_amd has no CPU fallback");
    if (opt.device < 0 || opt.device >= ndev) return fail(CVR_ERR_NO_DEVICE, "device %d out of range [0, %d)", opt.device, ndev);
    { const Chip chip = chip_of(opt.device); opt.cus = chip.cus; opt.xcds = chip.xcds; }
    Reader r;
    r.f = fopen(path, "rb");
    if (!r.f) return fail(CVR_ERR_IO, "cannot open %s", path);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{r.f};
    ImgKey have, want;
    r.pod(have);
    if (!r.ok || have.magic != kImgMagic) return fail(CVR_ERR_IO, "%s is not a CVR64 image file", path);
    make_key(want, expect, opt, have.vsz);
    if (memcmp(&have, &want, sizeof(have)) != 0) return fail(CVR_ERR_STATE, "%s was written for another source file, other options, another device geometry or library version", path);

    cvr_handle *h = new (std::nothrow) cvr_handle;
    if (!h) return fail(CVR_ERR_NOMEM, "out of host memory");
    h->device = opt.device; h->vsz = have.vsz; h->opt_used = opt;
    uint8_t *pinned = nullptr;
    const size_t pinned_bytes = (size_t)32 << 20;
#define LOAD_TRY(expr)                                                                                      \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess || !r.ok) {                                                                    \
            if (r.ok) fail(CVR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            else fail(CVR_ERR_IO, "%s is truncated or damaged", path);                                      \
            if (pinned) (void)hipHostFree(pinned);                                                          \
            cvr_destroy(h);                                                                                 \
            return r.ok ? CVR_ERR_HIP : CVR_ERR_IO;                                                         \
        }                                                                                                   \
    } while (0)
    LOAD_TRY(hipSetDevice(h->device));
    LOAD_TRY(acquire_stream(h->device, &h->stream));
    LOAD_TRY(hipHostMalloc(reinterpret_cast<void **>(&pinned), pinned_bytes, hipHostMallocDefault));
    h->events.resize(2);
    LOAD_TRY(hipEventCreate(&h->events[0]));
    LOAD_TRY(hipEventCreate(&h->events[1]));
    r.pod(h->info);
    uint32_t nparts = 0, nrounds = 0, has_multi = 0;
    r.pod(nparts); r.pod(h->ndict); r.pod(h->max_nshared); r.pod(h->multi_ystage); r.pod(nrounds); r.pod(has_multi);
    if (!r.ok || nparts == 0 || nparts > 64 || nrounds > 8 || (has_multi && nrounds != (nparts + 7) / 8) || (int64_t)nparts != (int64_t)std::max(h->info.col_panels, 1)) { r.ok = false; LOAD_TRY(hipSuccess); }
    std::vector<bool> slot_used((size_t)nrounds * 8, false);          // (has_multi: the parts' slots must be distinct)
    h->multi_chunks.resize(nrounds);
    for (uint32_t i = 0; i < nrounds; i++) r.pod(h->multi_chunks[i]);
    if (h->ndict > (uint32_t)cvr::kDictMax) { r.ok = false; LOAD_TRY(hipSuccess); }
    LOAD_TRY(r.dev(h->d_dict, pinned, pinned_bytes, h->stream, have.vsz * (uint64_t)cvr::kDictMax, true));
    if ((h->d_dict != nullptr) != (h->ndict != 0)) { r.ok = false; LOAD_TRY(hipSuccess); }
    h->parts.resize(nparts);
    const size_t vsz = h->vsz;
    int64_t      ztotal = 0;
    for (Part &p : h->parts) {
        ImgScalars s;
        r.pod(s);
        LOAD_TRY(hipSuccess);
        cvr::DeviceImage &g = p.img;
        g.S = s.S; g.G = s.G; g.f32 = s.f32 != 0; g.xcd_swizzle = s.xcd_swizzle; g.c16 = s.c16 != 0; g.tag16 = s.tag16 != 0;
        g.nchunks = s.nchunks; g.nrows = s.nrows; g.pad_col = s.pad_col; g.nshared = s.nshared; g.ystage = s.ystage; g.ndict = s.ndict;
        g.wpb = s.wpb; g.win_elems = s.win_elems; g.col_mask = s.col_mask; g.phases = s.phases; g.phase_width = s.phase_width; g.col_bits = s.col_bits;
        g.piece_max = s.piece_max; g.hub_n = s.hub_n; g.order_n = s.order_n; g.ncus = s.ncus;
        p.nrows = s.part_nrows; p.nnz = s.part_nnz; p.nnz_span = s.part_nnz_span; p.nchunks = s.part_nchunks; p.nshared = s.part_nshared; p.yext = s.part_yext; p.zoff = s.part_zoff;
        p.stream_bytes = (size_t)s.stream_bytes;
        p.multi_slot = s.multi_slot; g.ilv = s.ilv != 0; g.col_base = s.col_base; g.gang = s.gang;
        g.dict = h->d_dict;
        // the scalars must describe one consistent image: every array below is then required to have exactly the size they imply, and the
        // kernels' LDS and index arithmetic stays inside what cvr_create could have produced
        {
            const uint64_t nc = g.nchunks;
            const bool sane = s.S >= 4 && s.S <= 4096 && s.S % 4 == 0 && s.G == s.S / 4 && nc == (uint64_t)s.part_nchunks && g.nshared == (uint64_t)s.part_nshared && s.part_nrows >= 0 &&
                              (uint64_t)s.part_nrows == g.nrows && s.part_yext == s.part_nrows + 1 + 2 * (int64_t)nc && s.part_zoff >= 0 && g.wpb >= 1 && g.wpb <= (uint32_t)cvr::kMaxWavesPerBlock &&
                              g.phases >= 1 && g.phases <= 64 && g.ystage >= 1 && g.ystage <= 65532 && g.col_bits <= 31 && g.ndict == h->ndict && (g.f32 ? 4u : 8u) == have.vsz &&
                              (uint64_t)g.col_base + g.pad_col <= (uint64_t)h->info.ncols && (g.ilv ? h->info.col_panels > 1 || (g.col_base == 0 && g.pad_col == (uint64_t)h->info.ncols) : g.col_base == 0 && g.pad_col == (uint64_t)h->info.ncols) && (!g.c16 || (!g.tag16 && g.phases == 1)) && (g.order_n == 0 || g.order_n == g.pad_col) &&
                              s.stream_bytes == nc * (uint64_t)s.G * (uint64_t)cvr::group_bytes(g.f32, g.dict != nullptr, g.c16, g.tag16) &&
                              (s.multi_slot < 0 || (has_multi && (uint32_t)s.multi_slot < nrounds * 8)) && cvr::spmv_lds_bytes(g) <= cvr::kLdsBytes &&
                              (g.gang == 0 || (g.ilv && g.gang == g.wpb && g.gang >= 2 && (uint64_t)g.gang * g.ystage <= (g.tag16 ? 65536ull : 1ull << cvr::kGangTagBits) && (g.tag16 || g.col_bits == (uint32_t)cvr::kGangOffBits)));
            if (!sane) { r.ok = false; LOAD_TRY(hipSuccess); }
            // panels that run one per XCD: every part has a slot of its own in d_multi (two parts on one slot would leave one of them
            // unlaunched -- its slice of z stays zero, y silently wrong --; a part without a slot in a file that has the table likewise)
            if (has_multi) {
                if (s.multi_slot < 0 || slot_used[(size_t)s.multi_slot]) { r.ok = false; LOAD_TRY(hipSuccess); }
                slot_used[(size_t)s.multi_slot] = true;
            }
        }
        // (the stream allocation is padded for the kernel's run-ahead past the last chunk, as finish_part pads it)
        const size_t slack = 8 * (size_t)cvr::group_bytes(g.f32, g.dict != nullptr, g.c16, g.tag16);
        const uint64_t nc64 = g.nchunks;
        LOAD_TRY(r.dev(g.stream, pinned, pinned_bytes, h->stream, p.stream_bytes, false, p.stream_bytes + slack));
        if (!g.stream) LOAD_TRY(hipMalloc(&g.stream, std::max<size_t>(slack, 1)));
        LOAD_TRY(r.dev(g.desc, pinned, pinned_bytes, h->stream, 16 * nc64, false, 16));
        LOAD_TRY(r.dev(g.target, pinned, pinned_bytes, h->stream, 64 * nc64, false, 64));
        LOAD_TRY(r.dev(g.shared, pinned, pinned_bytes, h->stream, 24 * (uint64_t)g.nshared, false, 24));
        LOAD_TRY(r.dev(g.win_base, pinned, pinned_bytes, h->stream, sizeof(uint32_t) * (nc64 / g.wpb + 1)));
        LOAD_TRY(r.dev(g.desc2, pinned, pinned_bytes, h->stream, 8 * nc64, g.phases <= 1));
        if (g.phases > 1 && nc64 && !g.desc2) { r.ok = false; LOAD_TRY(hipSuccess); }
        LOAD_TRY(r.dev(g.cbase, pinned, pinned_bytes, h->stream, sizeof(uint32_t) * nc64, !g.c16));
        if (g.c16 && nc64 && !g.cbase) { r.ok = false; LOAD_TRY(hipSuccess); }
        LOAD_TRY(r.dev(g.gbase, pinned, pinned_bytes, h->stream, sizeof(uint32_t) * (nc64 * (uint64_t)g.G + 4096), !(g.gang && !g.tag16)));
        if (g.gang && !g.tag16 && !g.gbase) { r.ok = false; LOAD_TRY(hipSuccess); }
        LOAD_TRY(r.dev(g.hub_cols, pinned, pinned_bytes, h->stream, sizeof(int32_t) * (uint64_t)(g.order_n ? g.order_n : g.hub_n), g.hub_n == 0));
        if (g.hub_n && !g.hub_cols) { r.ok = false; LOAD_TRY(hipSuccess); }
        if (g.hub_n) LOAD_TRY(hipMalloc(&g.hub_x, vsz * (g.order_n ? ((size_t)g.order_n + 8) : ((g.hub_n + 3u) & ~3u))));
        ztotal = std::max<int64_t>(ztotal, p.zoff + p.yext);
    }
    if (h->paneled()) {
        const uint32_t nblocks = (uint32_t)((h->info.nrows + cvr::kCombineRows - 1) / cvr::kCombineRows);
        int64_t nsub_all = 0;
        for (const Part &p : h->parts) nsub_all += p.nrows;
        LOAD_TRY(r.dev(h->d_rows, pinned, pinned_bytes, h->stream, sizeof(uint32_t) * (uint64_t)std::max<int64_t>(nsub_all, 1)));
        LOAD_TRY(r.dev(h->d_block_off, pinned, pinned_bytes, h->stream, sizeof(uint32_t) * (uint64_t)nparts * (nblocks + 1)));
        LOAD_TRY(hipMalloc(&h->d_z, vsz * (size_t)std::max<int64_t>(ztotal, 1)));
        LOAD_TRY(hipMemsetAsync(h->d_z, 0, vsz * (size_t)std::max<int64_t>(ztotal, 1), h->stream));
        LOAD_TRY(hipMalloc(&h->d_rows16, sizeof(uint16_t) * (size_t)std::max<int64_t>(nsub_all, 1)));
        LOAD_TRY(cvr::launch_narrow_rows(h->d_rows, (size_t)nsub_all, h->d_rows16, h->stream));          // (what the combine pass reads: cvr_kernels.h, CombinePanel)
        std::vector<cvr::CombinePanel> cps(nparts);
        std::vector<cvr::FixPart>      fp(nparts);
        int64_t                        roff = 0;
        for (uint32_t i = 0; i < nparts; i++) {
            const Part &p = h->parts[i];
            uint8_t    *z = static_cast<uint8_t *>(h->d_z) + (size_t)p.zoff * vsz;
            cps[i] = cvr::CombinePanel{z, h->d_rows16 + roff};
            fp[i] = cvr::FixPart{p.img.shared, z, (uint32_t)p.nshared, (uint32_t)p.nrows};
            roff += p.nrows;
        }
        (void)nblocks;
        LOAD_TRY(hipMalloc(&h->d_cpanels, sizeof(cvr::CombinePanel) * nparts));
        LOAD_TRY(hipMalloc(&h->d_fixparts, sizeof(cvr::FixPart) * nparts));
        LOAD_TRY(hipMemcpy(h->d_cpanels, cps.data(), sizeof(cvr::CombinePanel) * nparts, hipMemcpyHostToDevice));
        LOAD_TRY(hipMemcpy(h->d_fixparts, fp.data(), sizeof(cvr::FixPart) * nparts, hipMemcpyHostToDevice));
        {          // the combine pass's bitmap is rebuilt from the row numbers (its bytes are in the saved image_bytes already)
            const int64_t bytes = h->info.image_bytes;
            if (setup_combine_bits(h, nsub_all) != CVR_OK) r.ok = false;
            h->info.image_bytes = bytes;
        }
        if (has_multi) {
            const size_t per_round = 8;
            std::vector<cvr::PanelArgs> pa((size_t)nrounds * 8, cvr::PanelArgs{nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u, nullptr, 0u, 0u});
            for (uint32_t j = 0; j < nparts; j++) {       // every panel where cvr_create placed it (the heaviest first, each on the XCD with the least work)
                const Part &p = h->parts[j];
                pa[p.multi_slot >= 0 ? (size_t)p.multi_slot : (j / per_round) * 8 + j % per_round] = cvr::PanelArgs{p.img.stream, p.img.desc, p.img.target, static_cast<uint8_t *>(h->d_z) + (size_t)p.zoff * vsz, p.img.nchunks, p.img.ystage, p.img.desc2, p.img.col_base, p.img.pad_col, p.img.gbase, 0u, 0u};
            }
            LOAD_TRY(hipMalloc(&h->d_multi, sizeof(cvr::PanelArgs) * pa.size()));
            LOAD_TRY(hipMemcpy(h->d_multi, pa.data(), sizeof(cvr::PanelArgs) * pa.size(), hipMemcpyHostToDevice));
        }
    }
    uint64_t tail = 0;
    r.pod(tail);
    if (tail != kImgMagic) r.ok = false;
    LOAD_TRY(hipSuccess);
    if (h->paneled() || h->info.hub_entries) LOAD_TRY(hipEventCreateWithFlags(&h->z_free, hipEventDisableTiming));
    LOAD_TRY(hipMalloc(&h->d_err, sizeof(uint32_t)));
    LOAD_TRY(hipMalloc(&h->d_x, vsz * (size_t)h->info.x_elems));
    LOAD_TRY(hipMalloc(&h->d_y, vsz * (size_t)std::max<int64_t>(h->info.yext_elems, 1)));
    LOAD_TRY(hipMemsetAsync(h->d_x, 0, vsz * (size_t)h->info.x_elems, h->stream));
    LOAD_TRY(hipMemsetAsync(h->d_y, 0, vsz * (size_t)std::max<int64_t>(h->info.yext_elems, 1), h->stream));
    LOAD_TRY(hipMemsetAsync(h->d_err, 0, sizeof(uint32_t), h->stream));
    LOAD_TRY(hipStreamSynchronize(h->stream));
    (void)hipHostFree(pinned);
    pinned = nullptr;
#undef LOAD_TRY
    { const int rcf = setup_fuse(h); if (rcf) { cvr_destroy(h); return rcf; } }          // (the fused combine's tables follow from the chunk tables: made again, not stored)
    ilv_runtime_settings(h);
    h->converted = true;
    // what the first run spent on analysis and conversion does not apply to this handle
    h->info.plan_s = 0; h->info.probe_s = 0; h->info.hub_select_s = 0; h->info.dict_s = 0; h->info.convert_s = 0; h->info.preprocess_wall_s = 0;
    h->info.upload_s = now_s() - t0;
    if (seconds) *seconds = now_s() - t0;
    *out = h;
    return CVR_OK;
}

}  // extern "C"

// cvr_panels.hip -- column panels from HOST arrays (the fallback of the device split, cvr_split.hip) and the panel rule: which share of the
// x gathers would miss a 4-MiB L2.
#include "cvr_internal.h"

using namespace cvrh;

// Column panels (the remedy when x outgrows the L2s).  The columns are cut into P ranges of
// equal width; panel p keeps, for every row that has a non-zero in its range, that row's entries of the
// range (rows compacted, order inside a row kept).  cmb_ptr / cmb_idx list, for every row, where its partial sums
// will stand in the concatenated y_ext buffers of the panels.
namespace cvrh {

// A parallel counting sort of the non-zeros by panel: row blocks are counted, then filled, by T host threads.
template <typename V>
void split_panels_t(const cvr_csr_view &v, int P, PanelSplit &out)
{
    const int64_t nrows = v.nrows, ncols = v.ncols, nz0 = nrows ? v.row_ptr[0] : 0, nz1 = nrows ? v.row_ptr[nrows] : 0;
    const V      *vals = static_cast<const V *>(v.vals);
    // panels are column ranges of equal width: what has to fit the L2 is the panel's slice of x, and the panels run
    // one after the other on the whole GPU, so their non-zero counts need not balance
    const int64_t width = (ncols + P - 1) / P > 0 ? (ncols + P - 1) / P : 1;
    auto          panel_of_col = [width](int32_t c) { return (int)(c / width); };
    int T = (int)std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if (T < 1 || nz1 - nz0 < (1 << 20)) T = 1;
    std::vector<int64_t> lo((size_t)T + 1);
    for (int t = 0; t <= T; t++) lo[(size_t)t] = nrows * t / T;
    // counts per (thread, panel): non-zeros and sub-rows
    std::vector<int64_t> cn((size_t)T * P, 0), cr((size_t)T * P, 0);
    auto count = [&](int t) {
        std::vector<int64_t> last((size_t)P, -1);
        for (int64_t r = lo[(size_t)t]; r < lo[(size_t)t + 1]; r++)
            for (int64_t j = v.row_ptr[r]; j < v.row_ptr[r + 1]; j++) {
                const int p = panel_of_col(v.col_idx[j]);
                cn[(size_t)t * P + p]++;
                if (last[(size_t)p] != r) { last[(size_t)p] = r; cr[(size_t)t * P + p]++; }
            }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(count, t);
        count(0);
        for (auto &x : th) x.join();
    }
    out.rp.resize((size_t)P); out.ci.resize((size_t)P); out.va.resize((size_t)P); out.rows.resize((size_t)P);
    std::vector<int64_t> on((size_t)T * P), orow((size_t)T * P);
    for (int p = 0; p < P; p++) {
        int64_t an = 0, ar = 0;
        for (int t = 0; t < T; t++) { on[(size_t)t * P + p] = an; orow[(size_t)t * P + p] = ar; an += cn[(size_t)t * P + p]; ar += cr[(size_t)t * P + p]; }
        out.ci[(size_t)p].alloc((size_t)an);
        out.va[(size_t)p].alloc((size_t)((an * (int64_t)sizeof(V) + 7) / 8));
        out.rows[(size_t)p].alloc((size_t)ar);
        out.rp[(size_t)p].alloc((size_t)ar + 1);
        out.rp[(size_t)p][(size_t)ar] = an;
    }
    auto fill = [&](int t) {
        std::vector<int64_t> last((size_t)P, -1), pn((size_t)P), pr((size_t)P);
        for (int p = 0; p < P; p++) { pn[(size_t)p] = on[(size_t)t * P + p]; pr[(size_t)p] = orow[(size_t)t * P + p]; }
        for (int64_t r = lo[(size_t)t]; r < lo[(size_t)t + 1]; r++)
            for (int64_t j = v.row_ptr[r]; j < v.row_ptr[r + 1]; j++) {
                const int p = panel_of_col(v.col_idx[j]);
                if (last[(size_t)p] != r) {
                    last[(size_t)p] = r;
                    out.rows[(size_t)p][(size_t)pr[(size_t)p]] = (uint32_t)r;
                    out.rp[(size_t)p][(size_t)pr[(size_t)p]] = pn[(size_t)p];
                    pr[(size_t)p]++;
                }
                out.ci[(size_t)p][(size_t)pn[(size_t)p]] = v.col_idx[j];
                reinterpret_cast<V *>(out.va[(size_t)p].data())[pn[(size_t)p]] = vals[j];
                pn[(size_t)p]++;
            }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(fill, t);
        fill(0);
        for (auto &x : th) x.join();
    }
}

// Which share of the x gathers would miss a 4-MiB L2?  Eight evenly spaced windows of 65 536 consecutive rows (what one
// XCD works on at a time is of that order): in each, the gathers are counted per 128-byte line of x; the 32 768 most
// used lines (4 MiB) are taken as resident, every other gather and every first touch of a line as a miss.  The windows
// are weighted by their non-zeros.  Banded matrices: ~0; R-MAT's hub columns keep it low (scale 22, fp64: 0.13) until the
// tail outgrows the cache (scale 24, fp32: 0.22); scattered columns with little re-use: 0.44 (LiveJournal shape).
// profiles/r01_panel_rule_l2_estimate.log
double l2_miss_estimate(const cvr_csr_view &v)
{
    const int64_t nrows = v.nrows, W = std::min<int64_t>(65536, nrows);
    if (W <= 0) return 0.0;
    const int64_t per_line = v.is_f32 ? 32 : 16, nlines = v.ncols / per_line + 1;
    const size_t  resident = (size_t)(4u << 20) / 128;
    const int     nwin = nrows == W ? 1 : 8;
    std::vector<double> refs_w((size_t)nwin, 0.0), miss_w((size_t)nwin, 0.0);
    // with the arrays on the device only the windows' slices of col_idx are fetched (row_ptr is a host copy by now)
    std::vector<std::vector<int32_t>> fetched((size_t)nwin);
    std::vector<const int32_t *>      base((size_t)nwin, nullptr);
    std::vector<int64_t>              shift((size_t)nwin, 0);
    for (int w = 0; w < nwin; w++) {
        const int64_t r0 = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
        const int64_t j0 = v.row_ptr[r0], j1 = v.row_ptr[r0 + W];
        if (v.arrays_on_device) {
            fetched[(size_t)w].resize((size_t)std::max<int64_t>(j1 - j0, 1));
            if (j1 > j0 && hipMemcpy(fetched[(size_t)w].data(), v.col_idx + j0, sizeof(int32_t) * (size_t)(j1 - j0), hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
            base[(size_t)w] = fetched[(size_t)w].data();
            shift[(size_t)w] = j0;
        } else {
            base[(size_t)w] = v.col_idx;
        }
    }
    auto window = [&](int w) {          // one thread per window, each with its own counters
        const int64_t r0 = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
        const int64_t j0 = v.row_ptr[r0], j1 = v.row_ptr[r0 + W];
        if (j1 <= j0) return;
        const int32_t *col = base[(size_t)w];
        const int64_t  sh = shift[(size_t)w];
        std::vector<uint32_t> cnt((size_t)nlines, 0u), touched;
        for (int64_t j = j0; j < j1; j++) {
            const size_t l = (size_t)(col[j - sh] / per_line);
            if (cnt[l]++ == 0) touched.push_back((uint32_t)l);
        }
        std::vector<uint32_t> top(touched.size());
        for (size_t i = 0; i < touched.size(); i++) top[i] = cnt[touched[i]];
        const size_t k = std::min(resident, top.size());
        if (k < top.size()) std::nth_element(top.begin(), top.begin() + (ptrdiff_t)k, top.end(), std::greater<uint32_t>());
        double hits = 0;
        for (size_t i = 0; i < k; i++) hits += (double)top[i] - 1.0;        // all but the first touch of a resident line
        refs_w[(size_t)w] = (double)(j1 - j0);
        miss_w[(size_t)w] = (double)(j1 - j0) - hits;
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nwin; w++) th.emplace_back(window, w);
    window(0);
    for (auto &t : th) t.join();
    double refs_all = 0, miss_all = 0;
    for (int w = 0; w < nwin; w++) { refs_all += refs_w[(size_t)w]; miss_all += miss_w[(size_t)w]; }
    return refs_all > 0 ? miss_all / refs_all : 0.0;
}

// the same estimate from a CSR in device memory (cvr_split.hip: l2_hits_device): same windows, same integers
hipError_t l2_miss_estimate_dev(const int64_t *rp_dev, const int32_t *ci_dev, int64_t nrows, int64_t ncols, bool f32, hipStream_t st, double *miss, cvr::Scratch lent)
{
    *miss = 0.0;
    const int64_t W = std::min<int64_t>(65536, nrows);
    if (W <= 0) return hipSuccess;
    const int nwin = nrows == W ? 1 : 8;
    int64_t   r0[8];
    double    refs[8], hits[8];
    for (int w = 0; w < nwin; w++) r0[w] = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
    const hipError_t e = cvr::l2_hits_device(rp_dev, ci_dev, r0, nwin, W, ncols, f32, (size_t)(4u << 20) / 128, refs, hits, st, lent);
    if (e != hipSuccess) return e;
    double refs_all = 0, miss_all = 0;
    for (int w = 0; w < nwin; w++) { refs_all += refs[w]; miss_all += refs[w] - hits[w]; }
    *miss = refs_all > 0 ? miss_all / refs_all : 0.0;
    return hipSuccess;
}

// The panel rule's second question.  Panels trade L2 misses of the gathers for partial sums: every (row, panel) pair with a non-zero is a
// value the panel kernel writes and the combine pass reads back with its row number.  pairs_per_nnz: those pairs per non-zero in the same
// eight windows of rows the miss estimate looks at, for panels of `width` columns.  Matrices of short rows whose misses are moderate lose
// with panels what they gain (round 5 hold-out, profiles/r05_holdout.log: a citation-like matrix of 4.5 non-zeros per row, miss 0.29,
// 0.59 pairs per non-zero ran 185 us as 16 panels, 139 us whole; a uniform random one, miss 0.65 / 0.44 pairs, 234 against 419; the
// web-Google shape x 2.2, 0.31 / 0.33, 55 against 72): panels_pay() keeps them from miss >= 0.68 pairs per non-zero on.
double pairs_per_nnz(const cvr_csr_view &v, int64_t width)
{
    const int64_t nrows = v.nrows, W = std::min<int64_t>(65536, nrows);
    if (W <= 0 || width < 1 || v.arrays_on_device) return 0.0;
    const int nwin = nrows == W ? 1 : 8;
    double    pairs = 0, refs = 0;
    for (int w = 0; w < nwin; w++) {
        const int64_t r0 = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
        for (int64_t r = r0; r < r0 + W; r++) {
            int64_t last = -1;
            for (int64_t j = v.row_ptr[r]; j < v.row_ptr[r + 1]; j++) {
                const int64_t p = v.col_idx[j] / width;
                if (p != last) { pairs += 1; last = p; }
            }
        }
        refs += (double)(v.row_ptr[r0 + W] - v.row_ptr[r0]);
    }
    return refs > 0 ? pairs / refs : 0.0;
}

hipError_t pairs_per_nnz_dev(const int64_t *rp_dev, const int32_t *ci_dev, int64_t nrows, int64_t width, hipStream_t st, double *out, cvr::Scratch lent)
{
    *out = 0.0;
    const int64_t W = std::min<int64_t>(65536, nrows);
    if (W <= 0 || width < 1) return hipSuccess;
    const int nwin = nrows == W ? 1 : 8;
    int64_t   r0[8];
    double    pairs[8], refs_w[8];
    for (int w = 0; w < nwin; w++) r0[w] = nwin == 1 ? 0 : (nrows - W) * w / (nwin - 1);
    hipError_t e = cvr::panel_pairs_device(rp_dev, ci_dev, r0, nwin, W, width, pairs, refs_w, st, lent);      // (the windows' non-zeros come back with the counts)
    if (e != hipSuccess) return e;
    double refs = 0, all = 0;
    for (int w = 0; w < nwin; w++) { refs += refs_w[w]; all += pairs[w]; }
    *out = refs > 0 ? all / refs : 0.0;
    return hipSuccess;
}

bool panels_pay(double miss, double pairs_per_nonzero) { return miss >= 0.68 * pairs_per_nonzero; }

// Thin lists: a gang's sorted list shares lines of x once it holds a few non-zeros per line of its panel's slice (requests per non-zero ~ (1 - exp(-d)) / d for
// d = non-zeros of a gang / lines of the slice); a matrix whose gangs get fewer than two takes panels half as wide when the partial sums that costs are few.
bool thin_lists(int P, double xbytes, int64_t nnz, int cus, double pairs_per_nonzero)
{
    if (P <= 1 || P > 32) return false;
    const double per_gang = std::min((double)nnz / (double)std::max(cus, 1), 100000.0);          // one gang per CU at least; ~100 000 non-zeros at most
    return per_gang < 2.0 * (xbytes / (double)P / 128.0) && pairs_per_nonzero < 0.15;
}

int panels_from_miss(double xb, double miss) { return miss > 0.17 ? std::min(64, std::max(2, (int)(xb * miss / 1.8e6 + 0.5))) : 1; }

// Panels that run one per XCD at a time, eight per launch (run_spmv, d_multi): the count the miss rule gave is for slices that share
// every L2 in turn; a slice with an L2 to itself may be ~2.6 MB, and the count is a multiple of eight.  CVR_XCD_PANELS=0 keeps the
// rule's count (each panel over the whole chip), CVR_XCD_PANELS=<n > 1> sets the count itself (experiments).
int xcd_panel_count(int P, double xbytes)
{
    const char *e = cvr::debug_env("xcd_panels");
    if (P <= 1 || (e && atoi(e) == 0)) return P;
    if (e && atoi(e) > 1) return std::min(64, atoi(e));
    if (xbytes <= 27e6) return 8;       // (the first round of eight stretches further: web-Google shapes of 23.5 / 26.4 MB 98.0 / 110.1 us as 8, 102.0 / 111.6 as 16 panels)
    return std::min(64, 8 * std::max(1, (int)std::ceil(xbytes / (8.0 * 2.6e6))));
}

int auto_panels(const cvr_csr_view &v, double *miss_out)
{
    const double xb = (double)v.ncols * (v.is_f32 ? 4.0 : 8.0);
    int          P = 1;
    double       miss = 0;
    const int64_t nnz = v.nrows > 0 ? v.row_ptr[v.nrows] - v.row_ptr[0] : 0;
    // (x of 12 .. 24 MB: only matrices beyond the resident layout of a whole MI355X; cvr_create also looks at the hub share there)
    if (xb >= 24e6 || (xb >= kMidPanelBytes && nnz > 0 && resident_out_of_reach(v.nrows, nnz, v.ncols, v.is_f32 != 0, make_iopt(nullptr)))) {
        miss = l2_miss_estimate(v);
        P = panels_from_miss(xb, miss);
    }
    if (miss_out) *miss_out = miss;
    P = xcd_panel_count(P, xb);         // (as cvr_create counts them on a whole MI355X: eight XCDs)
    if (P > 1 && !cvr::debug_env("no_pairs_rule")) {
        const double ppn = pairs_per_nnz(v, (v.ncols + P - 1) / P > 0 ? (v.ncols + P - 1) / P : 1);
        if (!panels_pay(miss, ppn)) P = 1;
        else if (thin_lists(P, xb, nnz, 256, ppn) && !cvr::debug_env("no_thin_lists_rule")) P *= 2;          // (cvr_create's rule for gang chunks, on a whole MI355X)
    }
    return P;
}

void split_panels(const cvr_csr_view &v, int P, PanelSplit &out)
{
    if (v.is_f32) split_panels_t<float>(v, P, out); else split_panels_t<double>(v, P, out);
}

}  // namespace cvrh

extern "C" {

int cvr_auto_panels(const cvr_csr_view *csr, double *l2_miss_estimate_out)
{
    cvr::debug_refresh();
    if (!csr) return fail(CVR_ERR_INVALID, "null argument");
    if (csr->arrays_on_device) return fail(CVR_ERR_INVALID, "cvr_auto_panels reads host arrays");
    if (csr->nrows > 0 && (!csr->row_ptr || (csr->row_ptr[csr->nrows] > 0 && !csr->col_idx))) return fail(CVR_ERR_INVALID, "null argument");
    return auto_panels(*csr, l2_miss_estimate_out);
}

}  // extern "C"

// cvr_mm.cpp -- Matrix-Market loader of the host program (readMatrix, /root/reference/spmv.cpp:311-535).
//
//   CVR_MM_REFCOMPAT reproduces the reference loader's arrays bit for bit, including its quirks
//                    (SURVEY.md Appendix B): indices stay 1-based (Q1), values pass through a float
//                    (Q2), pattern entries get value index%13 (Q3), only `symmetric` is mirrored
//                    (Q4), a last line without '\n' is dropped (Q5), nnz is padded to a multiple of
//                    16 with zero copies of the last entry (Q6), sorted by (row, col) (Q7), row
//                    pointers after the last non-empty row are nnz-1 (Q9).  Unlike the reference it
//                    uses 64-bit sizes (Q8) and returns errors instead of exit(1) (spmv.cpp:322-356).
//   CVR_MM_STRICT    is what the format means: 0-based, fp64 values (pattern = 1.0), symmetric /
//                    skew-symmetric / hermitian expansion, no padding.
//
// The reference parses with getline + sscanf on one thread and sorts 12-byte records with libc qsort
// (spmv.cpp:411-451, 485): seconds for web-Google, minutes for 10^8..10^9 entries.  Here the text is cut into
// one segment per OpenMP thread at line boundaries, every segment is parsed independently, and the entries
// are brought into (row, col) order by a stable counting sort on the row followed by a stable sort of each
// row -- the order libc's (merge-sort) qsort produces: entries with equal coordinates stay in file order.
// A binary image of the parsed CSR (cvr_mm_write_bin / cvr_mm_read_bin) skips the text altogether.
#include <omp.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cvr_amd.h"

namespace {

struct Rec { int64_t r; int32_t c; float vf; double vd; };   // vf: refcompat (through a float), vd: strict

struct Banner { bool pattern, complex_, symmetric, skew, hermitian; };

bool slurp(const char *path, std::vector<char> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.assign((size_t)(n > 0 ? n : 0) + 1, '\0');   // NUL-terminated
    const bool ok = n <= 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// next '\n'-terminated line of [p, end): std::getline(...).eof() as the reference loops on it
// (spmv.cpp:337, 377, 411) -- an unterminated last line does not count
bool next_line(const char *&p, const char *end, const char *&b, const char *&e)
{
    if (p >= end) return false;
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    if (!nl) { p = end; return false; }
    b = p; e = nl; p = nl + 1;
    return true;
}

int parse_banner(const char *&p, const char *end, Banner &b, long long &nr, long long &nc, long long &ne)
{
    const char *lb, *le;
    if (!next_line(p, end, lb, le)) return CVR_ERR_IO;
    std::string line(lb, le);
    char        id[128] = "", object[128] = "", format[128] = "", field[128] = "", symmetry[128] = "";
    sscanf(line.c_str(), "%127s %127s %127s %127s %127s", id, object, format, field, symmetry);
    if (strcmp(object, "matrix") != 0 || strcmp(format, "coordinate") != 0) return CVR_ERR_IO;   // spmv.cpp:346-356
    b.pattern = strcmp(field, "pattern") == 0;
    b.complex_ = strcmp(field, "complex") == 0;
    b.symmetric = strcmp(symmetry, "symmetric") == 0;
    b.skew = strcmp(symmetry, "skew-symmetric") == 0;
    b.hermitian = strcmp(symmetry, "hermitian") == 0;
    line.clear();
    while (next_line(p, end, lb, le)) {             // spmv.cpp:377-383
        line.assign(lb, le);
        if (line.empty() || line[0] != '%') break;
    }
    nr = nc = ne = 0;
    sscanf(line.c_str(), "%lld %lld %lld", &nr, &nc, &ne);   // spmv.cpp:386
    return CVR_OK;
}

inline bool is_ws(char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f' || ch == '\n'; }

// "%d"-like: skips white space, optional sign, digits; false (nothing stored) when no digit follows
inline bool scan_int(const char *&p, const char *e, long long &v)
{
    while (p < e && is_ws(*p)) p++;
    const char *q = p;
    bool        neg = false;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    if (q >= e || *q < '0' || *q > '9') return false;
    long long a = 0;
    while (q < e && *q >= '0' && *q <= '9') { a = a * 10 + (*q - '0'); q++; }
    v = neg ? -a : a;
    p = q;
    return true;
}

// "%f" (through a float, spmv.cpp:432) or a double, on a token bounded by the line
inline bool scan_real(const char *&p, const char *e, bool as_float, float &vf, double &vd)
{
    while (p < e && is_ws(*p)) p++;
    if (p >= e) return false;
    char   tok[96];
    size_t n = 0;
    while (p + n < e && !is_ws(p[n]) && n < sizeof(tok) - 1) { tok[n] = p[n]; n++; }
    tok[n] = 0;
    char *endp = nullptr;
    if (as_float) { vf = strtof(tok, &endp); vd = vf; } else { vd = strtod(tok, &endp); vf = (float)vd; }
    if (endp == tok) return false;
    p += (size_t)(endp - tok);
    return true;
}

struct Parsed {
    std::vector<std::vector<Rec>> part;    // entries of each text segment, file order, mirrors in place
    std::vector<long long>        offset;  // global index of each segment's first entry
    long long                     total = 0;
    int                           bad = 0;
};

// mode 0: the reference's line semantics (every '\n'-terminated line is an entry, fields that do not parse stay 0);
// mode 1: strict (blank and % lines skipped, 1-based indices checked and shifted to 0-based)
void parse_parallel(const char *p0, const char *end, const Banner &b, int mode, long long nRows, long long nCols, Parsed &out)
{
    int          T = omp_get_max_threads();
    const size_t len = (size_t)(end - p0);
    if (len < (1u << 16)) T = 1;
    std::vector<const char *> cut((size_t)T + 1);
    cut[0] = p0;
    cut[(size_t)T] = end;
    for (int t = 1; t < T; t++) {
        const char *q = p0 + len / (size_t)T * (size_t)t;
        const char *nl = static_cast<const char *>(memchr(q, '\n', (size_t)(end - q)));
        cut[(size_t)t] = nl ? nl + 1 : end;
    }
    for (int t = 1; t <= T; t++) if (cut[(size_t)t] < cut[(size_t)t - 1]) cut[(size_t)t] = cut[(size_t)t - 1];
    out.part.assign((size_t)T, {});
    out.offset.assign((size_t)T + 1, 0);
    const bool mirror = mode == 0 ? b.symmetric : (b.symmetric || b.skew || b.hermitian);
    int        bad = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(| : bad)
    for (int t = 0; t < T; t++) {
        std::vector<Rec> &v = out.part[(size_t)t];
        v.reserve((size_t)(cut[(size_t)t + 1] - cut[(size_t)t]) / 12 + 16);
        const char *p = cut[(size_t)t], *e = cut[(size_t)t + 1], *lb = nullptr, *le = nullptr;
        while (next_line(p, e, lb, le)) {
            const char *q = lb;
            if (mode == 1) {
                while (q < le && is_ws(*q)) q++;
                if (q == le || *q == '%') continue;
            }
            Rec       r{0, 0, 0.f, mode == 1 ? 1.0 : 0.0};
            long long a = 0, c = 0;
            bool      ok = scan_int(q, le, a);
            if (ok) { r.r = a; ok = scan_int(q, le, c); if (ok) r.c = (int32_t)c; }
            if (ok && !b.pattern) ok = scan_real(q, le, mode == 0, r.vf, r.vd);   // complex: real part only (spmv.cpp:423-428)
            if (mode == 1) {
                if (!ok || a < 1 || a > nRows || c < 1 || c > nCols) { bad |= 1; continue; }
                r.r = a - 1; r.c = (int32_t)(c - 1);
            }
            v.push_back(r);
            if (mirror && a != c) {                  // spmv.cpp:443-449
                Rec m = r;
                if (mode == 0) { m.r = c; m.c = (int32_t)a; }
                else { m.r = c - 1; m.c = (int32_t)(a - 1); if (b.skew) m.vd = -m.vd; }
                v.push_back(m);
            }
        }
    }
    for (int t = 0; t < T; t++) out.offset[(size_t)t + 1] = out.offset[(size_t)t] + (long long)out.part[(size_t)t].size();
    out.total = out.offset[(size_t)T];
    out.bad = bad;
}

// stable (row, col) order of recs with rows in [0, nrows): counting sort on the row, then each row by column
void sort_rows(std::vector<Rec> &recs, long long nrows, std::vector<int64_t> &rowstart)
{
    const size_t n = recs.size();
    rowstart.assign((size_t)nrows + 1, 0);
    for (size_t i = 0; i < n; i++) rowstart[(size_t)recs[i].r + 1]++;
    for (long long r = 0; r < nrows; r++) rowstart[(size_t)r + 1] += rowstart[(size_t)r];
    std::vector<Rec>     tmp(n);
    std::vector<int64_t> fill(rowstart.begin(), rowstart.end() - 1);
    for (size_t i = 0; i < n; i++) tmp[(size_t)fill[(size_t)recs[i].r]++] = recs[i];   // file order inside a row
    recs.swap(tmp);
#pragma omp parallel for schedule(dynamic, 4096)
    for (long long r = 0; r < nrows; r++) {
        Rec *a = recs.data() + rowstart[(size_t)r], *z = recs.data() + rowstart[(size_t)r + 1];
        bool sorted = true;
        for (Rec *t = a; t + 1 < z; t++) if (t[1].c < t[0].c) { sorted = false; break; }
        if (!sorted) std::stable_sort(a, z, [](const Rec &p, const Rec &q) { return p.c < q.c; });
    }
}

int read_refcompat(const char *p, const char *end, cvr_mm_matrix *out)
{
    Banner    b;
    long long nRows, nCols, nHdr;
    int       rc = parse_banner(p, end, b, nRows, nCols, nHdr);
    if (rc) return rc;
    if (nRows < 0 || nCols < 0) return CVR_ERR_INVALID;
    Parsed ps;
    parse_parallel(p, end, b, 0, nRows, nCols, ps);
    const long long index = ps.total;                // spmv.cpp:455
    if (index == 0) return CVR_ERR_IO;
    const long long npad = index % 16 == 0 ? index : (index + 16) / 16 * 16;   // spmv.cpp:457
    std::vector<Rec> recs((size_t)npad);
    const int        T = (int)ps.part.size();
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; t++) {
        // pattern files: value = running entry index % 13, the index counting mirrored entries too, a mirror
        // repeating its original's value (spmv.cpp:413-417, 443-449).  Inside a segment a mirror directly follows
        // its original and a segment starts with an original.
        long long g = ps.offset[(size_t)t];
        const std::vector<Rec> &v = ps.part[(size_t)t];
        for (size_t i = 0; i < v.size(); i++, g++) {
            Rec q = v[i];
            if (b.pattern) {
                q.vf = (float)(g % 13);
                recs[(size_t)g] = q;
                if (b.symmetric && q.r != q.c && i + 1 < v.size()) {   // its mirror
                    Rec m = v[i + 1];
                    m.vf = q.vf;
                    recs[(size_t)g + 1] = m;
                    i++; g++;
                }
            } else {
                recs[(size_t)g] = q;
            }
        }
    }
    ps.part.clear();
    const Rec last = recs[(size_t)index - 1];
    for (long long q = index; q < npad; q++) recs[(size_t)q] = Rec{last.r, last.c, 0.f, 0.0};   // spmv.cpp:474-482
    for (long long i = 0; i < npad; i++)
        if (recs[(size_t)i].r < 0 || recs[(size_t)i].r > nRows + 1 || recs[(size_t)i].c < 0) return CVR_ERR_INVALID;

    std::vector<int64_t> rowstart;
    sort_rows(recs, nRows + 2, rowstart);            // spmv.cpp:485 (rows 0 .. nRows+1 can occur)

    out->ref_numRows = nRows; out->ref_numCols = nCols; out->ref_nItems = npad; out->ref_nItemsRaw = index;
    out->row_ptr = static_cast<int64_t *>(malloc(sizeof(int64_t) * (size_t)(nRows + 2)));
    out->col_idx = static_cast<int32_t *>(malloc(sizeof(int32_t) * (size_t)npad));
    out->vals = static_cast<double *>(malloc(sizeof(double) * (size_t)npad));
    if (!out->row_ptr || !out->col_idx || !out->vals) return CVR_ERR_NOMEM;
    int32_t maxc = 0;
#pragma omp parallel for reduction(max : maxc)
    for (long long i = 0; i < npad; i++) {
        out->vals[i] = recs[(size_t)i].vf;
        out->col_idx[i] = recs[(size_t)i].c;
        if (recs[(size_t)i].c > maxc) maxc = recs[(size_t)i].c;
    }
    // the reference's row-pointer walk (spmv.cpp:499-526): rp[r] = first element of row r up to the last non-empty
    // row, nItems-1 afterwards (Q9)
    int64_t        *rp = out->row_ptr;
    const long long lastrow = recs[(size_t)npad - 1].r;
    for (long long r = 0; r <= nRows + 1; r++) rp[r] = r <= lastrow ? rowstart[(size_t)r] : npad - 1;
    // the arrays taken literally: rows 0..numRows (row 0 is always empty), columns 0..numCols
    out->nrows = nRows + 1;
    out->ncols = (nCols > maxc ? nCols : maxc) + 1;
    out->nnz = rp[nRows + 1];
    return CVR_OK;
}

int read_strict(const char *p, const char *end, cvr_mm_matrix *out)
{
    Banner    b;
    long long nRows, nCols, nHdr;
    int       rc = parse_banner(p, end, b, nRows, nCols, nHdr);
    if (rc) return rc;
    if (nRows < 0 || nCols < 0 || nCols >= 0x7fffffffLL) return CVR_ERR_INVALID;
    // strict mode also takes a last line without '\n': it is parsed from a copy that ends with one
    std::vector<char> tail;
    if (end > p && end[-1] != '\n') {
        const char *ls = end;
        while (ls > p && ls[-1] != '\n') ls--;
        tail.assign(ls, end);
        tail.push_back('\n');
        end = ls;
    }
    Parsed ps, pt;
    parse_parallel(p, end, b, 1, nRows, nCols, ps);
    if (!tail.empty()) parse_parallel(tail.data(), tail.data() + tail.size(), b, 1, nRows, nCols, pt);
    if (ps.bad || pt.bad) return CVR_ERR_INVALID;
    const long long  nnz = ps.total + pt.total;
    std::vector<Rec> recs((size_t)nnz);
    const int        T = (int)ps.part.size();
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; t++) std::copy(ps.part[(size_t)t].begin(), ps.part[(size_t)t].end(), recs.begin() + ps.offset[(size_t)t]);
    for (size_t t = 0; t < pt.part.size(); t++) std::copy(pt.part[t].begin(), pt.part[t].end(), recs.begin() + ps.total + pt.offset[t]);
    ps.part.clear();
    std::vector<int64_t> rowstart;
    sort_rows(recs, nRows, rowstart);
    out->ref_numRows = nRows; out->ref_numCols = nCols; out->ref_nItems = nnz; out->ref_nItemsRaw = nnz;
    out->nrows = nRows; out->ncols = nCols; out->nnz = nnz;
    out->row_ptr = static_cast<int64_t *>(malloc(sizeof(int64_t) * ((size_t)nRows + 1)));
    out->col_idx = static_cast<int32_t *>(malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1)));
    out->vals = static_cast<double *>(malloc(sizeof(double) * (size_t)(nnz ? nnz : 1)));
    if (!out->row_ptr || !out->col_idx || !out->vals) return CVR_ERR_NOMEM;
    std::copy(rowstart.begin(), rowstart.end(), out->row_ptr);
#pragma omp parallel for
    for (long long i = 0; i < nnz; i++) { out->col_idx[i] = recs[(size_t)i].c; out->vals[i] = recs[(size_t)i].vd; }
    return CVR_OK;
}

constexpr uint64_t kBinMagic = 0x3152534352564331ull;   // "1CVRCSR1"
constexpr uint64_t kBinMagicKeyed = 0x3252534352564332ull;   // "2CVRCSR2": the same image behind the key of its source file

uint64_t fnv1a(const unsigned char *p, size_t n, uint64_t h)
{
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

}  // namespace

extern "C" int cvr_mm_read(const char *path, int mode, cvr_mm_matrix *out)
{
    if (!path || !out) return CVR_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    std::vector<char> buf;
    if (!slurp(path, buf)) return CVR_ERR_IO;       // spmv.cpp:322-326
    const char *p = buf.data(), *end = buf.data() + buf.size() - 1;
    int         rc = mode == CVR_MM_REFCOMPAT ? read_refcompat(p, end, out) : read_strict(p, end, out);
    if (rc) cvr_mm_free(out);
    return rc;
}

extern "C" void cvr_mm_free(cvr_mm_matrix *m)
{
    if (!m) return;
    free(m->row_ptr); free(m->col_idx); free(m->vals);
    memset(m, 0, sizeof(*m));
}

// The identity of a source file, cheap to take: size, modification time, and a 64-bit FNV-1a hash over the first and the last MiB
// (a matrix edited in the middle without changing size or mtime escapes it; size and mtime catch what editors and downloads do).
extern "C" int cvr_source_key_of(const char *path, int mode, cvr_source_key *key)
{
    if (!path || !key) return CVR_ERR_INVALID;
    memset(key, 0, sizeof(*key));
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return CVR_ERR_IO;
    FILE *f = fopen(path, "rb");
    if (!f) return CVR_ERR_IO;
    const size_t               kMiB = (size_t)1 << 20;
    std::vector<unsigned char> buf(kMiB);
    uint64_t                   h = 0xcbf29ce484222325ull;
    size_t                     n = fread(buf.data(), 1, kMiB, f);
    h = fnv1a(buf.data(), n, h);
    if ((int64_t)st.st_size > (int64_t)kMiB) {
        const int64_t off = std::max<int64_t>((int64_t)kMiB, (int64_t)st.st_size - (int64_t)kMiB);
        if (fseeko(f, (off_t)off, SEEK_SET) == 0) { n = fread(buf.data(), 1, kMiB, f); h = fnv1a(buf.data(), n, h); }
    }
    fclose(f);
    key->size = (int64_t)st.st_size;
    key->mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    key->hash = h;
    key->mode = mode;
    return CVR_OK;
}

static int write_bin_impl(const char *path, const cvr_mm_matrix *m, const cvr_source_key *key);
static int read_bin_impl(const char *path, const cvr_source_key *expect, cvr_mm_matrix *out);

extern "C" int cvr_mm_write_bin_keyed(const char *path, const cvr_mm_matrix *m, const cvr_source_key *key)
{
    if (!key) return CVR_ERR_INVALID;
    return write_bin_impl(path, m, key);
}

extern "C" int cvr_mm_read_bin_keyed(const char *path, const cvr_source_key *expect, cvr_mm_matrix *out)
{
    if (!expect) return CVR_ERR_INVALID;
    return read_bin_impl(path, expect, out);
}

// readMatrix through the cache next to the file: <mtx>.ref.csrbin / <mtx>.strict.csrbin is read when its key is the file's, else
// the text is parsed and the cache rewritten (a cache that cannot be written is not an error)
extern "C" int cvr_mm_read_cached(const char *mtx_path, int mode, cvr_mm_matrix *out, int *cache_hit)
{
    if (cache_hit) *cache_hit = 0;
    if (!mtx_path || !out) return CVR_ERR_INVALID;
    cvr_source_key key;
    int            rc = cvr_source_key_of(mtx_path, mode, &key);
    if (rc) { memset(out, 0, sizeof(*out)); return rc; }
    const std::string bin = std::string(mtx_path) + (mode == CVR_MM_STRICT ? ".strict.csrbin" : ".ref.csrbin");
    if (read_bin_impl(bin.c_str(), &key, out) == CVR_OK) { if (cache_hit) *cache_hit = 1; return CVR_OK; }
    rc = cvr_mm_read(mtx_path, mode, out);
    if (rc == CVR_OK) (void)write_bin_impl(bin.c_str(), out, &key);
    return rc;
}

// Binary image of a parsed matrix: header {magic, 7 x int64}[, the source key], then row_ptr, col_idx, vals.
extern "C" int cvr_mm_write_bin(const char *path, const cvr_mm_matrix *m) { return write_bin_impl(path, m, nullptr); }

static int write_bin_impl(const char *path, const cvr_mm_matrix *m, const cvr_source_key *key)
{
    if (!path || !m || !m->row_ptr) return CVR_ERR_INVALID;
    const std::string tmp = std::string(path) + ".tmp";        // (written beside and renamed: a reader never sees half a file)
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return CVR_ERR_IO;
    const int64_t n = m->ref_nItems > m->nnz ? m->ref_nItems : m->nnz;   // refcompat keeps the excluded last element (Q9)
    const int64_t hdr[8] = {(int64_t)(key ? kBinMagicKeyed : kBinMagic), m->nrows, m->ncols, m->nnz, m->ref_numRows, m->ref_numCols, m->ref_nItems, m->ref_nItemsRaw};
    bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1;
    if (key) ok = ok && fwrite(key, sizeof(*key), 1, f) == 1;
    ok = ok && fwrite(m->row_ptr, sizeof(int64_t), (size_t)m->nrows + 1, f) == (size_t)m->nrows + 1;
    ok = ok && (n == 0 || fwrite(m->col_idx, sizeof(int32_t), (size_t)n, f) == (size_t)n);
    ok = ok && (n == 0 || fwrite(m->vals, sizeof(double), (size_t)n, f) == (size_t)n);
    ok = fclose(f) == 0 && ok;
    if (ok) ok = rename(tmp.c_str(), path) == 0;
    if (!ok) (void)remove(tmp.c_str());
    return ok ? CVR_OK : CVR_ERR_IO;
}

extern "C" int cvr_mm_read_bin(const char *path, cvr_mm_matrix *out) { return read_bin_impl(path, nullptr, out); }

// expect != null: only an image written with that very key is accepted (CVR_ERR_STATE for any other: stale or unkeyed)
static int read_bin_impl(const char *path, const cvr_source_key *expect, cvr_mm_matrix *out)
{
    if (!path || !out) return CVR_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    FILE *f = fopen(path, "rb");
    if (!f) return CVR_ERR_IO;
    int64_t hdr[8];
    if (fread(hdr, sizeof(hdr), 1, f) != 1 || ((uint64_t)hdr[0] != kBinMagic && (uint64_t)hdr[0] != kBinMagicKeyed) || hdr[1] < 0 || hdr[3] < 0 || hdr[6] < 0) { fclose(f); return CVR_ERR_IO; }
    if ((uint64_t)hdr[0] == kBinMagicKeyed) {
        cvr_source_key have;
        if (fread(&have, sizeof(have), 1, f) != 1) { fclose(f); return CVR_ERR_IO; }
        if (expect && memcmp(&have, expect, sizeof(have)) != 0) { fclose(f); return CVR_ERR_STATE; }      // the source has changed since
    } else if (expect) { fclose(f); return CVR_ERR_STATE; }
    out->nrows = hdr[1]; out->ncols = hdr[2]; out->nnz = hdr[3];
    out->ref_numRows = hdr[4]; out->ref_numCols = hdr[5]; out->ref_nItems = hdr[6]; out->ref_nItemsRaw = hdr[7];
    const int64_t n = out->ref_nItems > out->nnz ? out->ref_nItems : out->nnz;
    out->row_ptr = static_cast<int64_t *>(malloc(sizeof(int64_t) * ((size_t)out->nrows + 1)));
    out->col_idx = static_cast<int32_t *>(malloc(sizeof(int32_t) * (size_t)(n ? n : 1)));
    out->vals = static_cast<double *>(malloc(sizeof(double) * (size_t)(n ? n : 1)));
    bool ok = out->row_ptr && out->col_idx && out->vals;
    ok = ok && fread(out->row_ptr, sizeof(int64_t), (size_t)out->nrows + 1, f) == (size_t)out->nrows + 1;
    ok = ok && (n == 0 || fread(out->col_idx, sizeof(int32_t), (size_t)n, f) == (size_t)n);
    ok = ok && (n == 0 || fread(out->vals, sizeof(double), (size_t)n, f) == (size_t)n);
    fclose(f);
    if (!ok) { cvr_mm_free(out); return CVR_ERR_IO; }
    return CVR_OK;
}

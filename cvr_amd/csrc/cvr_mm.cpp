// cvr_mm.cpp -- Matrix-Market loader of the host program (readMatrix, /root/reference/spmv.cpp:311-535).
//
//   CVR_MM_REFCOMPAT reproduces the reference loader's arrays bit for bit, including its quirks
//                    (SURVEY.md Appendix B): indices stay 1-based (Q1), values pass through a float
//                    (Q2), pattern entries get value index%13 (Q3), only `symmetric` is mirrored
//                    (Q4), a last line without '\n' is dropped (Q5), nnz is padded to a multiple of
//                    16 with zero copies of the last entry (Q6), libc qsort on (row, col) (Q7), row
//                    pointers after the last non-empty row are nnz-1 (Q9).  Unlike the reference it
//                    uses 64-bit sizes (Q8) and returns errors instead of exit(1) (spmv.cpp:322-356).
//   CVR_MM_STRICT    is what the format means: 0-based, fp64 values (pattern = 1.0), symmetric /
//                    skew-symmetric / hermitian expansion, no padding, counting sort into CSR.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cvr_amd.h"

namespace {

struct Coord { int x, y; float val; };   // struct Coordinate, spmv.cpp:62-66

int coordcmp(const void *a, const void *b)   // spmv.cpp:131-144
{
    const Coord *p = static_cast<const Coord *>(a), *q = static_cast<const Coord *>(b);
    if (p->x != q->x) return p->x - q->x;
    return p->y - q->y;
}

// std::getline(...).eof() as the reference loops on it (spmv.cpp:337, 377, 411): only '\n'-terminated
// lines count
struct Lines {
    const char *buf; size_t len, pos;
    bool next(std::string &line)
    {
        if (pos >= len) return false;
        const void *nl = memchr(buf + pos, '\n', len - pos);
        if (!nl) { pos = len; return false; }
        const size_t n = (size_t)(static_cast<const char *>(nl) - (buf + pos));
        line.assign(buf + pos, n);
        pos += n + 1;
        return true;
    }
};

bool slurp(const char *path, std::vector<char> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.assign((size_t)(n > 0 ? n : 0) + 1, '\0');   // NUL-terminated for strtoll/strtod
    const bool ok = n <= 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

struct Banner { bool pattern, complex_, symmetric, skew, hermitian; };

int parse_banner(Lines &ln, Banner &b, long long &nr, long long &nc, long long &ne)
{
    std::string line;
    if (!ln.next(line)) return CVR_ERR_IO;
    char id[128] = "", object[128] = "", format[128] = "", field[128] = "", symmetry[128] = "";
    sscanf(line.c_str(), "%127s %127s %127s %127s %127s", id, object, format, field, symmetry);
    if (strcmp(object, "matrix") != 0 || strcmp(format, "coordinate") != 0) return CVR_ERR_IO;   // spmv.cpp:346-356
    b.pattern = strcmp(field, "pattern") == 0;
    b.complex_ = strcmp(field, "complex") == 0;
    b.symmetric = strcmp(symmetry, "symmetric") == 0;
    b.skew = strcmp(symmetry, "skew-symmetric") == 0;
    b.hermitian = strcmp(symmetry, "hermitian") == 0;
    line.clear();
    while (ln.next(line))                           // spmv.cpp:377-383
        if (line.empty() || line[0] != '%') break;
    nr = nc = ne = 0;
    sscanf(line.c_str(), "%lld %lld %lld", &nr, &nc, &ne);   // spmv.cpp:386
    return CVR_OK;
}

int read_refcompat(Lines &ln, cvr_mm_matrix *out)
{
    Banner    b;
    long long nRows, nCols, nHdr;
    int       rc = parse_banner(ln, b, nRows, nCols, nHdr);
    if (rc) return rc;
    std::vector<Coord> co;
    co.reserve((size_t)(nHdr > 0 ? nHdr : 16) * (b.symmetric ? 2 : 1) + 32);
    std::string line;
    long long   index = 0;
    while (ln.next(line)) {                         // spmv.cpp:411-451
        Coord c{0, 0, 0.f};
        if (b.pattern) {
            sscanf(line.c_str(), "%d %d", &c.x, &c.y);
            c.val = (float)(index % 13);            // spmv.cpp:417
        } else if (b.complex_) {
            float im;
            sscanf(line.c_str(), "%d %d %f %f", &c.x, &c.y, &c.val, &im);
        } else {
            sscanf(line.c_str(), "%d %d %f", &c.x, &c.y, &c.val);   // spmv.cpp:432 (through a float)
        }
        co.push_back(c);
        index++;
        if (b.symmetric && c.x != c.y) {            // spmv.cpp:443-449
            co.push_back(Coord{c.y, c.x, c.val});
            index++;
        }
    }
    if (index == 0) return CVR_ERR_IO;
    const long long npad = index % 16 == 0 ? index : (index + 16) / 16 * 16;   // spmv.cpp:457
    const Coord     last = co.back();
    for (long long q = index; q < npad; q++) co.push_back(Coord{last.x, last.y, 0.f});   // spmv.cpp:474-482
    qsort(co.data(), (size_t)npad, sizeof(Coord), coordcmp);                          // spmv.cpp:485
    for (long long i = 0; i < npad; i++)
        if (co[(size_t)i].x < 0 || co[(size_t)i].x > nRows + 1 || co[(size_t)i].y < 0) return CVR_ERR_INVALID;

    out->ref_numRows = nRows; out->ref_numCols = nCols; out->ref_nItems = npad; out->ref_nItemsRaw = index;
    out->row_ptr = static_cast<int64_t *>(malloc(sizeof(int64_t) * (size_t)(nRows + 2)));
    out->col_idx = static_cast<int32_t *>(malloc(sizeof(int32_t) * (size_t)npad));
    out->vals = static_cast<double *>(malloc(sizeof(double) * (size_t)npad));
    if (!out->row_ptr || !out->col_idx || !out->vals) return CVR_ERR_NOMEM;
    int64_t *rp = out->row_ptr;
    rp[0] = 0;                                      // spmv.cpp:499
    long long r = 0, i = 0;
    int32_t   maxc = 0;
    for (; i < npad; i++) {                         // spmv.cpp:505-514
        while (co[(size_t)i].x != r) rp[++r] = i;
        out->vals[i] = co[(size_t)i].val;
        out->col_idx[i] = co[(size_t)i].y;
        if (co[(size_t)i].y > maxc) maxc = co[(size_t)i].y;
    }
    for (long long k = r + 1; k <= nRows + 1; k++) rp[k] = i - 1;   // spmv.cpp:522-526 (Q9)
    // the arrays taken literally: rows 0..numRows (row 0 is always empty), columns 0..numCols
    out->nrows = nRows + 1;
    out->ncols = (nCols > maxc ? nCols : maxc) + 1;
    out->nnz = rp[nRows + 1];
    return CVR_OK;
}

struct Entry { int64_t r; int32_t c; double v; };

int read_strict(Lines &ln, cvr_mm_matrix *out)
{
    Banner    b;
    long long nRows, nCols, nHdr;
    int       rc = parse_banner(ln, b, nRows, nCols, nHdr);
    if (rc) return rc;
    if (nRows < 0 || nCols < 0 || nCols >= 0x7fffffffLL) return CVR_ERR_INVALID;
    std::vector<Entry> en;
    en.reserve((size_t)(nHdr > 0 ? nHdr : 16) * ((b.symmetric || b.skew || b.hermitian) ? 2 : 1));
    // the rest of the buffer, including a last line without '\n'
    const char *p = ln.buf + ln.pos, *end = ln.buf + ln.len;
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) p++;
        if (p >= end) break;
        if (*p == '%') { while (p < end && *p != '\n') p++; continue; }
        char     *q;
        long long r = strtoll(p, &q, 10);
        if (q == p) return CVR_ERR_IO;
        p = q;
        long long c = strtoll(p, &q, 10);
        if (q == p) return CVR_ERR_IO;
        p = q;
        double v = 1.0;
        if (!b.pattern) {
            v = strtod(p, &q);
            if (q == p) return CVR_ERR_IO;
            p = q;
            if (b.complex_) { (void)strtod(p, &q); p = q; }   // real part only, as the reference (spmv.cpp:423-428)
        }
        while (p < end && *p != '\n') p++;
        if (r < 1 || r > nRows || c < 1 || c > nCols) return CVR_ERR_INVALID;
        en.push_back(Entry{r - 1, (int32_t)(c - 1), v});
        if ((b.symmetric || b.skew || b.hermitian) && r != c) en.push_back(Entry{c - 1, (int32_t)(r - 1), b.skew ? -v : v});
    }
    const size_t nnz = en.size();
    out->ref_numRows = nRows; out->ref_numCols = nCols; out->ref_nItems = (int64_t)nnz; out->ref_nItemsRaw = (int64_t)nnz;
    out->nrows = nRows; out->ncols = nCols; out->nnz = (int64_t)nnz;
    out->row_ptr = static_cast<int64_t *>(calloc((size_t)nRows + 1, sizeof(int64_t)));
    out->col_idx = static_cast<int32_t *>(malloc(sizeof(int32_t) * (nnz ? nnz : 1)));
    out->vals = static_cast<double *>(malloc(sizeof(double) * (nnz ? nnz : 1)));
    if (!out->row_ptr || !out->col_idx || !out->vals) return CVR_ERR_NOMEM;
    // counting sort by row, then (stable) by column inside each row
    int64_t *rp = out->row_ptr;
    for (const Entry &e : en) rp[e.r + 1]++;
    for (long long r = 0; r < nRows; r++) rp[r + 1] += rp[r];
    std::vector<int64_t> fill(rp, rp + nRows);
    std::vector<Entry>   byrow(nnz);
    for (const Entry &e : en) byrow[(size_t)fill[(size_t)e.r]++] = e;
    for (long long r = 0; r < nRows; r++) {
        Entry *a = byrow.data() + rp[r], *z = byrow.data() + rp[r + 1];
        bool   sorted = true;
        for (Entry *t = a; t + 1 < z; t++) if (t[1].c < t[0].c) { sorted = false; break; }
        if (!sorted) std::stable_sort(a, z, [](const Entry &p, const Entry &q) { return p.c < q.c; });   // file order among duplicates
    }
    for (size_t i = 0; i < nnz; i++) { out->col_idx[i] = byrow[i].c; out->vals[i] = byrow[i].v; }
    return CVR_OK;
}

}  // namespace

extern "C" int cvr_mm_read(const char *path, int mode, cvr_mm_matrix *out)
{
    if (!path || !out) return CVR_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    std::vector<char> buf;
    if (!slurp(path, buf)) return CVR_ERR_IO;       // spmv.cpp:322-326
    Lines ln{buf.data(), buf.size() - 1, 0};
    int   rc = mode == CVR_MM_REFCOMPAT ? read_refcompat(ln, out) : read_strict(ln, out);
    if (rc) cvr_mm_free(out);
    return rc;
}

extern "C" void cvr_mm_free(cvr_mm_matrix *m)
{
    if (!m) return;
    free(m->row_ptr); free(m->col_idx); free(m->vals);
    memset(m, 0, sizeof(*m));
}

/* oracle/cvr_oracle.c -- TEST INFRASTRUCTURE.  NOT part of the product (see cvr_oracle.h).
 *
 * CPU restatement, in plain C, of the reference's path in /root/reference/spmv.cpp.  Every function
 * cites the reference lines it follows.  Checked against the unmodified reference's outputs in
 * tests/golden/NAME.npz (tests/test_oracle_golden.py): loader arrays and CSR y bit-for-bit, the
 * 8-lane CVR arrays bit-for-bit, CVR y within the stated fp64 tolerance of the CSR y.
 */
#define _GNU_SOURCE
#include "cvr_oracle.h"

#include <limits.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Loader: readMatrix, spmv.cpp:311-535
 * ---------------------------------------------------------------------------------------------- */
struct coord { int x, y; float val; };              /* struct Coordinate, spmv.cpp:62-66 */

static int coordcmp(const void *v1, const void *v2)  /* spmv.cpp:131-144 */
{
    const struct coord *c1 = (const struct coord *)v1, *c2 = (const struct coord *)v2;
    if (c1->x != c2->x) return c1->x - c2->x;
    return c1->y - c2->y;
}

/* std::getline(...).eof() semantics of spmv.cpp:337, 377, 411: a line counts only when it is
 * terminated by '\n'; a final unterminated line sets eof and is DROPPED. */
static int next_line(const char *buf, size_t len, size_t *pos, char *line, size_t cap)
{
    size_t p = *pos;
    if (p >= len) return 0;
    const char *nl = memchr(buf + p, '\n', len - p);
    if (!nl) { *pos = len; return 0; }
    size_t n = (size_t)(nl - (buf + p));
    if (n >= cap) n = cap - 1;
    memcpy(line, buf + p, n);
    line[n] = 0;
    *pos = (size_t)(nl - buf) + 1;
    return 1;
}

int orc_read_matrix(const char *path, orc_csr *out)
{
    memset(out, 0, sizeof(*out));
    FILE *f = fopen(path, "rb");
    if (!f) return -1;                                  /* spmv.cpp:322-326 */
    fseek(f, 0, SEEK_END);
    long flen = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)flen + 1);
    if (!buf || fread(buf, 1, (size_t)flen, f) != (size_t)flen) { fclose(f); free(buf); return -1; }
    fclose(f);
    size_t len = (size_t)flen, pos = 0;
    enum { LCAP = 4096 };
    char line[LCAP];
    char id[128] = "", object[128] = "", format[128] = "", field[128] = "", symmetry[128] = "";

    if (!next_line(buf, len, &pos, line, LCAP)) { free(buf); return -2; }       /* spmv.cpp:337 */
    sscanf(line, "%127s %127s %127s %127s %127s", id, object, format, field, symmetry);
    if (strcmp(object, "matrix") != 0) { free(buf); return -2; }                /* spmv.cpp:346 */
    if (strcmp(format, "coordinate") != 0) { free(buf); return -3; }            /* spmv.cpp:352 */
    int pattern = strcmp(field, "pattern") == 0;                                /* spmv.cpp:358 */
    int field_complex = strcmp(field, "complex") == 0;                          /* spmv.cpp:363 */
    int symmetric = strcmp(symmetry, "symmetric") == 0;                         /* spmv.cpp:368 */

    line[0] = 0;
    while (next_line(buf, len, &pos, line, LCAP))                               /* spmv.cpp:377-383 */
        if (line[0] != '%') break;
    int nRows = 0, nCols = 0, nElements = 0;
    sscanf(line, "%d %d %d", &nRows, &nCols, &nElements);                       /* spmv.cpp:386 */

    /* the reference sizes its buffer from the header (spmv.cpp:390-402, int overflow Q8) and never
     * checks it; here the buffer grows on demand */
    size_t cap = (size_t)(nElements > 0 ? nElements : 16) * (symmetric ? 2 : 1) + 32;
    struct coord *co = (struct coord *)calloc(cap, sizeof(*co));
    int index = 0;
    while (next_line(buf, len, &pos, line, LCAP)) {                             /* spmv.cpp:411-451 */
        if ((size_t)index + 34 > cap) {
            cap *= 2;
            co = (struct coord *)realloc(co, cap * sizeof(*co));
            memset(co + index, 0, (cap - (size_t)index) * sizeof(*co));
        }
        if (pattern) {
            sscanf(line, "%d %d", &co[index].x, &co[index].y);
            co[index].val = (float)(index % 13);                                /* spmv.cpp:417, Q3 */
        } else if (field_complex) {
            float im;
            sscanf(line, "%d %d %f %f", &co[index].x, &co[index].y, &co[index].val, &im); /* :426 */
        } else {
            sscanf(line, "%d %d %f", &co[index].x, &co[index].y, &co[index].val); /* spmv.cpp:432, Q2 */
        }
        index++;                                        /* indices stay 1-based: spmv.cpp:437-438 */
        if (symmetric && co[index - 1].x != co[index - 1].y) {                  /* spmv.cpp:443-449 */
            co[index].x = co[index - 1].y;
            co[index].y = co[index - 1].x;
            co[index].val = co[index - 1].val;
            index++;
        }
    }
    free(buf);
    nElements = index;                                                          /* spmv.cpp:455 */
    if (nElements == 0) { free(co); return -4; }
    int npad = (nElements % 16 == 0) ? nElements : (nElements + 16) / 16 * 16;  /* spmv.cpp:457 */
    for (int q = index; q < npad; q++) {                                        /* spmv.cpp:474-482, Q6 */
        co[q].x = co[index - 1].x;
        co[q].y = co[index - 1].y;
        co[q].val = 0;
    }
    qsort(co, (size_t)npad, sizeof(struct coord), coordcmp);                    /* spmv.cpp:485, Q7 */

    for (int i = 0; i < npad; i++)
        if (co[i].x < 0 || co[i].x > nRows + 1) { free(co); return -5; } /* reference: heap overrun */

    out->nItems = npad;
    out->nItemsRaw = nElements;
    out->numRows = nRows;
    out->numCols = nCols;
    out->val = (double *)malloc(sizeof(double) * (size_t)npad);
    out->cols = (int *)malloc(sizeof(int) * (size_t)npad);
    out->rowptr = (int *)malloc(sizeof(int) * ((size_t)nRows + 2));
    int *rp = out->rowptr;
    rp[0] = 0;                                                                  /* spmv.cpp:499 */
    int r = 0, i;
    for (i = 0; i < npad; i++) {                                                /* spmv.cpp:505-514 */
        while (co[i].x != r) rp[++r] = i;
        out->val[i] = co[i].val;
        out->cols[i] = co[i].y;
    }
    for (int k = r + 1; k <= nRows + 1; k++) rp[k] = i - 1;                     /* spmv.cpp:522-526, Q9 */
    free(co);
    return 0;
}

void orc_free_csr(orc_csr *m)
{
    free(m->val); free(m->cols); free(m->rowptr);
    memset(m, 0, sizeof(*m));
}

/* ------------------------------------------------------------------------------------------------
 * CSR oracle: spmv.cpp:1843-1850
 * ---------------------------------------------------------------------------------------------- */
void orc_csr_spmv(int numRows, const int *rowptr, const int *cols, const double *val,
                  const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < numRows; i++) {
        double sum = 0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; j++) sum += val[j] * x[cols[j]];
        y[i] = sum;
    }
}

void orc_csr_spmv64(int64_t nrows, const int64_t *rowptr, const int32_t *cols, const double *val,
                    const double *x, double *y, double *absy)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nrows; i++) {
        double sum = 0, a = 0;
        for (int64_t j = rowptr[i]; j < rowptr[i + 1]; j++) {
            double p = val[j] * x[cols[j]];
            sum += p;
            a += fabs(p);
        }
        y[i] = sum;
        if (absy) absy[i] = a;
    }
}

/* fp32 data, fp64 accumulation: the oracle for the fp32 device path (no reference counterpart) */
void orc_csr_spmv64_f32(int64_t nrows, const int64_t *rowptr, const int32_t *cols, const float *val,
                        const float *x, double *y, double *absy)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nrows; i++) {
        double sum = 0, a = 0;
        for (int64_t j = rowptr[i]; j < rowptr[i + 1]; j++) {
            double p = (double)val[j] * (double)x[cols[j]];
            sum += p;
            a += fabs(p);
        }
        y[i] = sum;
        if (absy) absy[i] = a;
    }
}

double orc_x_rand(uint64_t j)
{
    uint64_t z = 0xC0FFEEull + (j + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

/* ------------------------------------------------------------------------------------------------
 * CSR -> 8-lane CVR: pre_processing, spmv.cpp:565-1014 (format: SURVEY Appendix A)
 * ---------------------------------------------------------------------------------------------- */
/* The reference reads rowDelimiters[numRows+2] (one past its allocation, spmv.cpp:687, 837) when a
 * chunk ends in trailing empty rows; the value is undefined there.  Here such a read returns a
 * value that makes the "row is empty" test false, which is what any non-matching heap word does. */
static inline int RP(const orc_csr *m, int k)
{
    return k <= m->numRows + 1 ? m->rowptr[k] : m->rowptr[m->numRows + 1] + 1;
}

int orc_cvr8_convert(const orc_csr *m, int T, orc_cvr8 *c)
{
    memset(c, 0, sizeof(*c));
    const int nItems = m->nItems, numRows = m->numRows;
    if (T < 1 || nItems / T / 16 < 1) return -1;        /* the reference crashes on empty chunks */
    c->T = T; c->nItems = nItems; c->numRows = numRows;
    c->vals = (double *)malloc(sizeof(double) * (size_t)nItems);
    c->cols = (int *)malloc(sizeof(int) * (size_t)nItems);
    c->record_len = 2 * ((int64_t)numRows + 240 + (int64_t)T * 32);              /* spmv.cpp:1806 */
    c->record = (int *)malloc(sizeof(int) * (size_t)c->record_len);
    for (int64_t i = 0; i < c->record_len; i++) c->record[i] = ORC_RECORD_SENTINEL;
    c->split = (int *)calloc((size_t)2 * T, sizeof(int));                        /* spmv.cpp:1813-1820 */
    c->final2 = (int *)malloc(sizeof(int) * 16 * (size_t)T);
    for (int i = 0; i < 16 * T; i++) c->final2[i] = ORC_RECORD_SENTINEL;
    c->nnz_rows = (int *)calloc((size_t)4 * T, sizeof(int));

    const int thread_nnz = (nItems / T / 16) * 16;                               /* spmv.cpp:584 */
    const int thread_break = (nItems - thread_nnz * T) / 16;                     /* spmv.cpp:585 */

    for (int t = 0; t < T; t++) {   /* reference: one OpenMP thread each; chunks are independent */
        int seg = 0;
        int s, e;
        if (t < thread_break) { s = t * (thread_nnz + 16); e = (t + 1) * (thread_nnz + 16); } /* :615 */
        else { s = t * thread_nnz + thread_break * 16; e = (t + 1) * thread_nnz + thread_break * 16; }
        if (t == T - 1) e = nItems;                                              /* spmv.cpp:626 */

        int start = 0, stop = numRows, median;                                   /* spmv.cpp:631-650 */
        while (stop >= start) {
            median = (stop + start) / 2;
            if (s >= m->rowptr[median]) start = median + 1; else stop = median - 1;
        }
        int rows_start = start - 1;
        start = rows_start; stop = numRows;                                      /* spmv.cpp:652-667 */
        while (stop >= start) {
            median = (stop + start) / 2;
            if ((e - 1) >= m->rowptr[median]) start = median + 1; else stop = median - 1;
        }
        int rows_end = start - 1;
        while ((RP(m, rows_end + 1) - RP(m, rows_end) == 0) && rows_end <= numRows) rows_end++; /* :687 */
        c->nnz_rows[4 * t] = s; c->nnz_rows[4 * t + 1] = e;                      /* spmv.cpp:690-694 */
        c->nnz_rows[4 * t + 2] = rows_start; c->nnz_rows[4 * t + 3] = rows_end;
        const int rows_span = rows_end - rows_start + 1;

        double *ov = c->vals + s; int *oc = c->cols + s;
        const double *iv = m->val + s; const int *ic = m->cols + s;
        int *fin2 = c->final2 + t * 16;
        int *rec = c->record + 2 * (t * 32 + rows_start) / 16 * 16;              /* spmv.cpp:709 */

        int valID[8], rowID[8], count[8], flag[16], first_flag[8];
        int count_hi_zero = 0;      /* upper 8 SIMD lanes of `count`: 1 before step 0, 0 after (:980) */
        for (int k = 0; k < 16; k++) flag[k] = -1;
        for (int k = 0; k < 8; k++) first_flag[k] = 0;
        const int rows_start_init = rows_start;
        int rs = rows_start;
        for (int i = 0; i < 8; i++) {                                            /* spmv.cpp:727-759 */
            if (rs < rows_end) {
                valID[i] = RP(m, rs) - s; rowID[i] = rs; count[i] = RP(m, rs + 1) - RP(m, rs);
            } else if (rs == rows_end) {
                valID[i] = RP(m, rs) - s; rowID[i] = rs; count[i] = e - RP(m, rs);
            } else { valID[i] = 0; rowID[i] = 0; count[i] = 0; }
            if (i == 0) {
                valID[i] = 0;
                count[i] = RP(m, rs + 1) - s;
                if (rs == rows_end) count[i] = e - s;
            }
            rs++;
        }
        int first_in = 0;
        const int steps = (e - s) / 8;
        for (int i = 0; i < steps; i++) {                                        /* spmv.cpp:808 */
            int mn = count[0];
            for (int k = 1; k < 8; k++) if (count[k] < mn) mn = count[k];
            if (mn == 0) {                                                       /* spmv.cpp:810-946 */
                for (int kk = 0; kk < 8; kk++) {
                    if (count[kk] != 0) continue;
                    if (rs <= rows_end) {                                        /* feeding, :821-868 */
                        if (rowID[kk] == rows_start_init) c->split[2 * t] = i * 8 + kk;
                        else { rec[seg] = i * 8 + kk; rec[seg + 1] = rowID[kk]; seg += 2; }
                        while (RP(m, rs + 1) - RP(m, rs) == 0) rs++;             /* :837-838 */
                        valID[kk] = RP(m, rs) - s; rowID[kk] = rs; count[kk] = RP(m, rs + 1) - RP(m, rs);
                        if (rs == rows_end) {                                    /* :844-857 */
                            if (c->split[2 * t + 1] == 0) c->split[2 * t + 1] = i * 8 + kk;
                            count[kk] = e - RP(m, rs);
                            for (int q = 0; q < 8; q++) fin2[q] = rowID[q];
                            for (int q = 0; q < 8; q++) if (count[q] == 0) flag[q] = 0;
                            if (count_hi_zero) for (int q = 8; q < 16; q++) flag[q] = 0;
                        }
                        rs++;
                    } else {                                                     /* stealing, :869-943 */
                        int sum = 0;
                        for (int q = 0; q < 8; q++) sum += count[q];
                        const int ave = sum / 8;                                 /* :871 */
                        int cand = 0;
                        for (cand = 0; cand < 8; cand++) if (count[cand] > ave) break; /* :876-879 */
                        if (first_flag[kk] == 0) {
                            if (first_in == 0) {
                                if (c->split[2 * t + 1] == 0)
                                    c->split[2 * t + 1] = (rows_span <= 8) ? -1 : i * 8 + kk; /* :885-891 */
                                for (int q = 0; q < 8; q++) fin2[q] = rowID[q];  /* :893 */
                                first_in = 1;
                            }
                            rec[seg] = i * 8 + kk; rec[seg + 1] = kk;            /* :898-899 */
                            flag[kk] = cand; first_flag[kk] = 1;
                        } else {                                                 /* dead code, A.6 */
                            rec[seg] = i * 8 + kk; rec[seg + 1] = flag[kk];
                            flag[kk] = cand;
                        }
                        if (cand < 8) {                                          /* :927-931 */
                            valID[kk] = valID[cand]; rowID[kk] = cand; count[kk] = ave;
                            count[cand] -= ave; valID[cand] += ave;
                        }
                        seg += 2;
                    }
                }
            }
            for (int k = 0; k < 8; k++) {                                        /* spmv.cpp:963-980 */
                ov[i * 8 + k] = iv[valID[k]];
                oc[i * 8 + k] = ic[valID[k]];
                valID[k]++; count[k]--;
            }
            count_hi_zero = 1;
            if (i == steps - 1)                                                  /* spmv.cpp:982-999 */
                for (int kk = 0; kk < 8; kk++) {
                    rec[seg] = -1; rec[seg + 1] = (flag[kk] == -1) ? kk : flag[kk]; seg += 2;
                }
        }
    }
    return 0;
}

void orc_cvr8_free(orc_cvr8 *c)
{
    free(c->vals); free(c->cols); free(c->record); free(c->split); free(c->final2); free(c->nnz_rows);
    memset(c, 0, sizeof(*c));
}

/* ------------------------------------------------------------------------------------------------
 * 8-lane CVR SpMV: the semantics of spmv_compute_kernel, spmv.cpp:1016-1667, written
 * as ONE step loop instead of the five hand-split phases so that
 *   K1 (steal records misread as row stores when ncsr == -1, spmv.cpp:1245, 1286-1295) and
 *   K2 (a step executed by both phase B and phase D, spmv.cpp:1245, 1351, 1450)
 * cannot occur.  A record (pos, wb) with pos <= ncsr is a row store (spmv.cpp:1489-1498), a later
 * one adds into the staging slot t_result[wb] (spmv.cpp:1525-1545, 1607-1616); the chunk's first
 * row is added atomically at ncsr_start (spmv.cpp:1280-1282); the tail folds lanes into t_result and
 * adds it atomically into y[final_2[k]] (spmv.cpp:1631-1651).
 * ---------------------------------------------------------------------------------------------- */
void orc_cvr8_spmv(const orc_cvr8 *c, const double *x, double *y, int nthreads)
{
    const int T = c->T;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < c->numRows + 2; i++) y[i] = 0;                           /* spmv.cpp:1026-1031 */

#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < T; t++) {
        const int s = c->nnz_rows[4 * t], e = c->nnz_rows[4 * t + 1];
        const int first_row = c->nnz_rows[4 * t + 2];
        const double *cv = c->vals + s; const int *cc = c->cols + s;
        const int *rec = c->record + 2 * (t * 32 + first_row) / 16 * 16;         /* spmv.cpp:1112 */
        const int *fin2 = c->final2 + t * 16;
        const int ncsr_start = c->split[2 * t], ncsr = c->split[2 * t + 1];      /* spmv.cpp:1153-1154 */
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tres[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int ri = 0;
        const int steps = (e - s) / 8;
        for (int i = 0; i < steps; i++) {
            if (ncsr_start != 0 && (ncsr_start >> 3) == i) {                     /* spmv.cpp:1280-1282 */
                const int l = ncsr_start & 7;
#pragma omp atomic
                y[first_row] += acc[l];
                acc[l] = 0;
            }
            while (rec[ri] >= 0 && (rec[ri] >> 3) == i) {                        /* spmv.cpp:1197-1224 */
                const int l = rec[ri] & 7, wb = rec[ri + 1];
                if (rec[ri] <= ncsr) y[wb] = acc[l];                             /* exclusive owner */
                else tres[wb] += acc[l];
                acc[l] = 0;
                ri += 2;
            }
            const double *v = cv + (size_t)i * 8; const int *cl = cc + (size_t)i * 8;
            for (int l = 0; l < 8; l++) acc[l] = fma(v[l], x[cl[l]], acc[l]);    /* _mm512_fmadd_pd, spmv.cpp:1226-1233 */
        }
        for (int k = 0; k < 8; k++) { tres[rec[ri + 1]] += acc[k]; ri += 2; }    /* spmv.cpp:1633-1638 */
        for (int k = 0; k < 8; k++) {                                            /* spmv.cpp:1640-1649 */
            int w = fin2[k];
            /* final_2 is only written at the last feed / first steal (spmv.cpp:853, 893); a chunk of
             * <= 8 equal-length rows has neither and the reference then reads an unset word.  The
             * lanes still hold their initial rows first_row + k (spmv.cpp:727-759). */
            if (w == ORC_RECORD_SENTINEL) { w = first_row + k; if (w > c->nnz_rows[4 * t + 3]) continue; }
#pragma omp atomic
            y[w] += tres[k];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * bench.py helper: write a 0-based CSR pattern as a row-major `pattern general` Matrix-Market file, so
 * that the unmodified reference binary (oracle/_ref/spmv.cvr.ref) can be timed on the bench matrix.
 * ---------------------------------------------------------------------------------------------- */
int orc_write_mtx_pattern(const char *path, int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *ci)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    static char buf[1 << 20];
    setvbuf(f, buf, _IOFBF, sizeof(buf));
    fprintf(f, "%%%%MatrixMarket matrix coordinate pattern general\n%lld %lld %lld\n", (long long)nrows,
            (long long)ncols, (long long)(rp[nrows] - rp[0]));
    for (int64_t r = 0; r < nrows; r++)
        for (int64_t j = rp[r]; j < rp[r + 1]; j++) fprintf(f, "%lld %d\n", (long long)(r + 1), ci[j] + 1);
    return fclose(f) ? -1 : 0;
}

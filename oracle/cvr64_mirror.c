/* oracle/cvr64_mirror.c -- TEST INFRASTRUCTURE.  NOT part of the product (see cvr_oracle.h).
 *
 * CPU mirror of the repo's 64-lane device format "CVR64" (DESIGN.md section 3).  It exists so that
 *   (1) the HIP converter (cvr_amd/csrc/cvr_convert.hip) can be checked bit-for-bit, and
 *   (2) the format + write-back rules can be validated against the CSR oracle without a GPU.
 * CVR64 re-derives the reference's CVR idea (pre_processing, spmv.cpp:565-1014: W concurrent
 * trackers, greedy "next non-empty row to the first free lane" feeding, spmv.cpp:821-868, and
 * "steal `ave` elements from the first over-full lane", spmv.cpp:869-943) for W = 64 lanes:
 *   - chunks are cut at row boundaries (rows longer than a threshold are split), all chunks have
 *     exactly S steps, the remainder is a zero "pad row" (the reference pads nnz instead, :474-482)
 *   - the end of a lane segment is bit 31 of the column word (replaces the (pos,wb) record list,
 *     spmv.cpp:832-834, 898-899); rows are handed out by rank among the finishing lanes
 *   - the steal part keeps per-lane staging slots (t_result, spmv.cpp:1607-1616) via `target`
 *   - rows shared by several chunks go to carry slots + an ordered fix-up (replaces the fp64
 *     atomics of spmv.cpp:1280-1282, 1640-1649), so y needs no zeroing
 */
#include "cvr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define W 64

/* largest index j in [lo, hi] with rowptr[j] <= key (rowptr non-decreasing, rowptr[lo] <= key) */
static int64_t last_le(const int64_t *rp, int64_t lo, int64_t hi, int64_t key)
{
    while (lo < hi) {
        int64_t mid = lo + (hi - lo + 1) / 2;
        if (rp[mid] <= key) lo = mid; else hi = mid - 1;
    }
    return lo;
}

/* ---- planner: chunk boundaries (host side in the product too: cvr_plan.cpp) ---- */
static int64_t plan(int64_t nrows, const int64_t *rp, int64_t cap, int64_t thr, int64_t **nzb_out,
                    int64_t **rf_out, int64_t **rl_out)
{
    const int64_t nnz = rp[nrows];
    int64_t capn = 16, n = 0;
    int64_t *nzb = (int64_t *)malloc(sizeof(int64_t) * (size_t)(capn + 1));
    int64_t *rf = (int64_t *)malloc(sizeof(int64_t) * (size_t)capn);
    int64_t *rl = (int64_t *)malloc(sizeof(int64_t) * (size_t)capn);
    int64_t pos = 0;
    while (pos < nnz) {
        if (n == capn) {
            capn *= 2;
            nzb = (int64_t *)realloc(nzb, sizeof(int64_t) * (size_t)(capn + 1));
            rf = (int64_t *)realloc(rf, sizeof(int64_t) * (size_t)capn);
            rl = (int64_t *)realloc(rl, sizeof(int64_t) * (size_t)capn);
        }
        const int64_t begin = pos, limit = begin + cap;
        /* row holding element `begin`: last r with rp[r] <= begin (then rp[r+1] > begin) */
        const int64_t r0 = last_le(rp, 0, nrows, begin);
        /* rows r0..j-1 end at or before `limit` */
        int64_t j = last_le(rp, r0, nrows, limit);
        pos = rp[j];
        if (pos < begin) pos = begin;               /* j == r0: not even the first row fits */
        if (j < nrows && pos < limit) {
            const int64_t rem = rp[j + 1] - pos;    /* row j does not fit in limit - pos */
            if (rem > thr) pos = limit;             /* long row: split it, fill the chunk */
        }
        nzb[n] = begin;
        rf[n] = r0;
        rl[n] = last_le(rp, r0, nrows, pos - 1);    /* row holding element pos-1 */
        n++;
    }
    nzb[n] = nnz;
    *nzb_out = nzb; *rf_out = rf; *rl_out = rl;
    return n;
}

int orc_cvr64_build(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols,
                    const void *vals, int is_f32, int S, int64_t thr, orc_cvr64 *c)
{
    memset(c, 0, sizeof(*c));
    if (S < 4 || S % 4) return -1;
    const int64_t cap = (int64_t)W * S;
    if (thr <= 0 || thr > cap) thr = cap / 4;
    c->nrows = nrows; c->ncols = ncols; c->nnz = rp[nrows]; c->S = S; c->is_f32 = is_f32;
    c->nchunks = plan(nrows, rp, cap, thr, &c->nz_begin, &c->row_first, &c->row_last);
    const int64_t NC = c->nchunks;
    const int G = S / 4;
    c->cols = (uint32_t *)calloc((size_t)(NC * cap) + 4, sizeof(uint32_t));
    c->vals = calloc((size_t)(NC * cap) + 4, is_f32 ? 4 : 8);
    c->desc = (uint32_t *)calloc((size_t)NC * 2 + 2, sizeof(uint32_t));
    c->target = (uint8_t *)calloc((size_t)NC * W + 1, 1);

    /* pass 1: rows per chunk (non-empty rows in [row_first,row_last] + pad row), shared rows */
    int64_t nseg = 0, nshared = 0;
    for (int64_t k = 0; k < NC; k++) {
        int64_t n = 0;
        for (int64_t r = c->row_first[k]; r <= c->row_last[k]; r++) n += rp[r + 1] > rp[r];
        if (c->nz_begin[k + 1] - c->nz_begin[k] < cap) n++;
        c->desc[2 * k] = (uint32_t)nseg;
        c->desc[2 * k + 1] = (uint32_t)n;
        nseg += n;
        if (c->nz_begin[k + 1] < rp[c->row_last[k] + 1] && c->nz_begin[k] <= rp[c->row_last[k]])
            nshared++;                                /* a row that starts here and continues */
    }
    c->nseg = nseg;
    c->dest = (uint32_t *)calloc((size_t)nseg + 1, sizeof(uint32_t));
    c->nshared = nshared;
    c->shared_row = (int64_t *)calloc((size_t)nshared + 1, sizeof(int64_t));
    c->shared_c0 = (int64_t *)calloc((size_t)nshared + 1, sizeof(int64_t));
    c->shared_c1 = (int64_t *)calloc((size_t)nshared + 1, sizeof(int64_t));
    int64_t ns = 0;
    for (int64_t k = 0; k < NC; k++) {
        const int64_t rl = c->row_last[k];
        if (c->nz_begin[k + 1] < rp[rl + 1] && c->nz_begin[k] <= rp[rl]) {
            int64_t k1 = k + 1;
            while (c->nz_begin[k1 + 1] < rp[rl + 1]) k1++;     /* last chunk holding row rl */
            c->shared_row[ns] = rl; c->shared_c0[ns] = k; c->shared_c1[ns] = k1; ns++;
        }
    }

    /* pass 2: per-chunk tracker simulation */
    int64_t *seg_start = NULL; int64_t *seg_cnt = NULL; int64_t segcap = 0;
    for (int64_t k = 0; k < NC; k++) {
        const int64_t b = c->nz_begin[k], e = c->nz_begin[k + 1], L = e - b;
        const int64_t rbase = c->desc[2 * k];
        const int64_t n = c->desc[2 * k + 1];
        if (n > segcap) {
            segcap = n * 2;
            seg_start = (int64_t *)realloc(seg_start, sizeof(int64_t) * (size_t)segcap);
            seg_cnt = (int64_t *)realloc(seg_cnt, sizeof(int64_t) * (size_t)segcap);
        }
        /* the chunk's row list: (relative start, count, destination) */
        int64_t q = 0;
        for (int64_t r = c->row_first[k]; r <= c->row_last[k]; r++) {
            if (rp[r + 1] == rp[r]) continue;
            const int64_t s0 = rp[r] > b ? rp[r] : b, e0 = rp[r + 1] < e ? rp[r + 1] : e;
            seg_start[q] = s0 - b; seg_cnt[q] = e0 - s0;
            uint32_t d = (uint32_t)r;
            if (rp[r] < b) d = (uint32_t)(nrows + 1 + 2 * k);            /* head shared  */
            else if (rp[r + 1] > e) d = (uint32_t)(nrows + 1 + 2 * k + 1); /* tail shared */
            c->dest[rbase + q] = d;
            q++;
        }
        if (L < cap) { seg_start[q] = L; seg_cnt[q] = cap - L; c->dest[rbase + q] = (uint32_t)nrows; q++; }
        if (q != n) { fprintf(stderr, "cvr64 mirror: row count mismatch\n"); return -2; }

        int64_t pos[W], cnt[W];
        int64_t fed = 0;
        for (int l = 0; l < W; l++) { pos[l] = 0; cnt[l] = 0; c->target[k * W + l] = (uint8_t)l; }
        for (int i = 0; i < S; i++) {
            /* refill: empty lanes in lane order; feed while rows remain, then steal */
            const int64_t ave = S - i;
            for (int l = 0; l < W; l++) {
                if (cnt[l] != 0) continue;
                if (fed < n) { pos[l] = seg_start[fed]; cnt[l] = seg_cnt[fed]; fed++; }
                else {
                    int v = 0;
                    while (v < W && cnt[v] <= ave) v++;      /* first over-full lane (spmv.cpp:876-879) */
                    if (v == W) { fprintf(stderr, "cvr64 mirror: no victim\n"); return -3; }
                    pos[l] = pos[v]; cnt[l] = ave;           /* stealer takes the FIRST ave (spmv.cpp:927-931) */
                    pos[v] += ave; cnt[v] -= ave;
                    c->target[k * W + l] = (uint8_t)v;
                }
            }
            const int g = i / 4, j = i % 4;
            for (int l = 0; l < W; l++) {
                uint32_t col = 0; double v = 0;
                if (pos[l] < L) {
                    col = (uint32_t)cols[b + pos[l]];
                    v = is_f32 ? (double)((const float *)vals)[b + pos[l]] : ((const double *)vals)[b + pos[l]];
                }
                if (cnt[l] == 1) col |= 0x80000000u;
                const size_t ci = (((size_t)k * G + g) * W + l) * 4 + j;
                c->cols[ci] = col;
                if (is_f32) ((float *)c->vals)[ci] = (float)v;
                else {
                    const int h = j / 2, jj = j % 2;
                    ((double *)c->vals)[((((size_t)k * G + g) * 2 + h) * W + l) * 2 + jj] = v;
                }
                pos[l]++; cnt[l]--;
            }
        }
        for (int l = 0; l < W; l++)
            if (cnt[l] != 0) { fprintf(stderr, "cvr64 mirror: lane not drained\n"); return -4; }
    }
    free(seg_start); free(seg_cnt);
    return 0;
}

void orc_cvr64_free(orc_cvr64 *c)
{
    free(c->nz_begin); free(c->row_first); free(c->row_last); free(c->cols); free(c->vals);
    free(c->desc); free(c->dest); free(c->target); free(c->shared_row); free(c->shared_c0);
    free(c->shared_c1);
    memset(c, 0, sizeof(*c));
}

/* Interpret a CVR64 image with the HIP kernel's rules (cvr_amd/csrc/cvr_spmv.hip):
 *   FEED lane, flagged, rows remain      -> y_ext[cur] = acc; take row fed+rank
 *   FEED lane, flagged, rows exhausted   -> y_ext[cur] = acc; lane turns stealer (no own tail row)
 *   FEED lane, flagged, after feeding ended (tail mode) -> own = acc; lane turns stealer
 *   end of chunk: slot[lane] = own (or acc if still FEED); slot[target[lane]] += stolen acc;
 *                 y_ext[cur] = slot[lane] for lanes that own a tail row
 *   fix-up: y[row] = carry(tail of c0) + sum of carry(head of c) for c0 < c <= c1, in chunk order */
void orc_cvr64_spmv(const orc_cvr64 *c, const void *xv, void *yv)
{
    const int S = c->S, G = S / 4;
    const int64_t NC = c->nchunks, nrows = c->nrows;
    const size_t next = (size_t)(nrows + 1 + 2 * NC);
    double *yext = (double *)calloc(next, sizeof(double));
    for (int64_t k = 0; k < NC; k++) {
        const uint32_t rbase = c->desc[2 * k], n = c->desc[2 * k + 1];
        double acc[W], own[W];
        uint32_t cur[W];
        int mode_feed[W], has_own[W];
        uint32_t fed = n < W ? n : W;
        for (int l = 0; l < W; l++) {
            acc[l] = 0; own[l] = 0;
            mode_feed[l] = (uint32_t)l < fed; has_own[l] = mode_feed[l];
            cur[l] = mode_feed[l] ? c->dest[rbase + l] : 0;
        }
        int tail = fed == n;
        for (int i = 0; i < S; i++) {
            const int g = i / 4, j = i % 4;
            int flagged[W];
            for (int l = 0; l < W; l++) {
                const size_t ci = (((size_t)k * G + g) * W + l) * 4 + j;
                const uint32_t cw = c->cols[ci];
                const uint32_t col = cw & 0x7fffffffu;
                flagged[l] = cw >> 31;
                if (c->is_f32) {
                    const float v = ((const float *)c->vals)[ci];
                    acc[l] = (double)fmaf(v, ((const float *)xv)[col], (float)acc[l]);
                } else {
                    const double v = ((const double *)c->vals)[((((size_t)k * G + g) * 2 + j / 2) * W + l) * 2 + j % 2];
                    acc[l] = fma(v, ((const double *)xv)[col], acc[l]);
                }
            }
            if (i == S - 1) break;
            if (!tail) {
                uint32_t rank = 0, nf = 0;
                for (int l = 0; l < W; l++) {
                    if (!flagged[l]) continue;
                    yext[cur[l]] = acc[l]; acc[l] = 0;
                    if (fed + rank < n) { cur[l] = c->dest[rbase + fed + rank]; nf++; }
                    else { mode_feed[l] = 0; has_own[l] = 0; }
                    rank++;
                }
                fed += nf;
                tail = fed == n;
            } else {
                for (int l = 0; l < W; l++)
                    if (flagged[l] && mode_feed[l]) { own[l] = acc[l]; acc[l] = 0; mode_feed[l] = 0; }
            }
        }
        double slot[W];
        for (int l = 0; l < W; l++) {
            if (mode_feed[l]) { own[l] = acc[l]; acc[l] = 0; }
            slot[l] = has_own[l] ? own[l] : 0;
        }
        for (int l = 0; l < W; l++)
            if (!mode_feed[l]) {
                const int t = c->target[k * W + l];
                if (c->is_f32) slot[t] = (double)((float)slot[t] + (float)acc[l]); else slot[t] += acc[l];
            }
        for (int l = 0; l < W; l++) if (has_own[l]) yext[cur[l]] = slot[l];
    }
    for (int64_t s = 0; s < c->nshared; s++) {
        double v = yext[nrows + 1 + 2 * c->shared_c0[s] + 1];
        for (int64_t k = c->shared_c0[s] + 1; k <= c->shared_c1[s]; k++) {
            if (c->is_f32) v = (double)((float)v + (float)yext[nrows + 1 + 2 * k]);
            else v += yext[nrows + 1 + 2 * k];
        }
        yext[c->shared_row[s]] = v;
    }
    if (c->is_f32) for (int64_t r = 0; r < nrows; r++) ((float *)yv)[r] = (float)yext[r];
    else memcpy(yv, yext, sizeof(double) * (size_t)nrows);
    free(yext);
}

/* oracle/cvr64_mirror.c -- TEST INFRASTRUCTURE.  NOT part of the product (see cvr_oracle.h).
 *
 * CPU mirror of the repo's 64-lane device format "CVR64" (DESIGN.md section 3).  It exists so that
 *   (1) the HIP converter (cvr_amd/csrc/cvr_convert.hip) can be checked bit-for-bit, and
 *   (2) the format + write-back rules can be validated against the CSR oracle without a GPU.
 * It is written from the format's definition with the reference's SEQUENTIAL semantics (lanes
 * served one by one in lane order, spmv.cpp:814-946), not from the wave-parallel device code, so an
 * equal image means the ballot/rank formulation on the device hands out rows exactly like the
 * reference's scalar loop does.
 *
 * CVR64 re-derives the reference's CVR idea (pre_processing, spmv.cpp:565-1014: W concurrent
 * trackers, greedy "next row to the first free lane" feeding, spmv.cpp:821-868, and "steal `ave`
 * elements from the first over-full lane", spmv.cpp:869-943) for W = 64 lanes:
 *   - every row owns max(1, nnz) slots (an empty row owns one pad slot: column = ncols, value 0),
 *     so the row a lane writes follows from the hand-out order and no (pos, wb) records exist
 *     (spmv.cpp:832-834, 898-899)
 *   - chunks are cut at row boundaries (only rows longer than a threshold are cut), every chunk
 *     has exactly 64*S slots, the remainder is a pad segment (the reference pads nnz, :474-482)
 *   - the end of a lane segment is bit 31 of the column word
 *   - the steal part keeps per-lane staging slots (t_result, spmv.cpp:1607-1616) via `target`
 *   - rows cut over several chunks go to carry slots + an ordered fix-up (replaces the fp64
 *     atomics of spmv.cpp:1280-1282, 1640-1649), so y needs no zeroing
 */
#include "cvr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define W 64
#define PLAN_ROW_BLOCK 65536

typedef struct { int64_t nzb, row_first, nrows_in, nseg, pad; int head, tail; } chunk_t;

/* planner: a sequential walk over the rows (the product's is cvr_amd/csrc/cvr_plan.cpp) */
static int64_t plan(int64_t nrows, const int64_t *rp, int64_t cap, int64_t thr, int64_t max_rows, chunk_t **out,
                    int64_t **sh_out, int64_t *nsh_out)
{
    if (max_rows <= 0) max_rows = INT64_MAX;
    int64_t capn = 16, n = 0, shcap = 16, nsh = 0;
    chunk_t *ch = (chunk_t *)malloc(sizeof(chunk_t) * (size_t)capn);
    int64_t *sh = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)shcap);
    int64_t r = 0, off = 0;
    while (r < nrows) {
        if (n == capn) { capn *= 2; ch = (chunk_t *)realloc(ch, sizeof(chunk_t) * (size_t)capn); }
        chunk_t c;
        c.row_first = r; c.nzb = rp[r] + off; c.head = off > 0;
        int64_t used = 0;
        /* the walk restarts at every multiple of 65536 rows (the product plans these blocks in parallel): the chunk ends there */
        const int64_t block_end = (r / PLAN_ROW_BLOCK + 1) * PLAN_ROW_BLOCK < nrows ? (r / PLAN_ROW_BLOCK + 1) * PLAN_ROW_BLOCK : nrows;
        while (r < block_end) {
            if (r - c.row_first >= max_rows) break;        /* row cap (column phases): the rest is padding */
            const int64_t len = rp[r + 1] - rp[r] - off;
            const int64_t slots = len > 0 ? len : 1;
            if (used + slots <= cap) { used += slots; r++; off = 0; if (used == cap) break; continue; }
            if (slots > thr) { off += cap - used; used = cap; }
            break;
        }
        c.tail = off > 0;
        const int64_t last = c.tail ? r : r - 1;
        c.nrows_in = last - c.row_first + 1;
        c.pad = cap - used;
        c.nseg = c.nrows_in + (c.pad > 0);
        if (c.head && !(c.tail && last == c.row_first)) sh[3 * (nsh - 1) + 2] = n;
        if (c.tail && !(c.head && last == c.row_first)) {
            if (nsh == shcap) { shcap *= 2; sh = (int64_t *)realloc(sh, sizeof(int64_t) * 3 * (size_t)shcap); }
            sh[3 * nsh] = last; sh[3 * nsh + 1] = n; sh[3 * nsh + 2] = -1; nsh++;
        }
        ch[n++] = c;
    }
    *out = ch; *sh_out = sh; *nsh_out = nsh;
    return n;
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* value dictionary: the distinct bit patterns of the values plus +0.0 (pad slots), sorted; -1 if more than 256 */
static int build_dict(const void *vals, int is_f32, int64_t n0, int64_t n1, uint64_t *dict)
{
    int n = 0;
    dict[n++] = 0;
    for (int64_t j = n0; j < n1; j++) {
        const uint64_t b = is_f32 ? (uint64_t)((const uint32_t *)vals)[j] : ((const uint64_t *)vals)[j];
        int k = 0;
        while (k < n && dict[k] != b) k++;
        if (k == n) { if (n == 256) return -1; dict[n++] = b; }
    }
    qsort(dict, (size_t)n, sizeof(uint64_t), cmp_u64);
    return n;
}

int orc_cvr64_build(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols,
                    const void *vals, int is_f32, int S, int64_t thr, orc_cvr64 *c)
{
    return orc_cvr64_build_dict(nrows, ncols, rp, cols, vals, is_f32, S, thr, 0, c);
}

int orc_cvr64_build_dict(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols,
                         const void *vals, int is_f32, int S, int64_t thr, int use_dict, orc_cvr64 *c)
{
    return orc_cvr64_build_ex(nrows, ncols, rp, cols, vals, is_f32, S, thr, use_dict, 1, 0, c);
}

/* phases > 1: "column phases".  The columns are cut into `phases` ranges of equal width (a multiple of 16); a chunk feeds
 * its rows' pieces phase by phase -- every (row, phase) pair with a non-zero is one segment, in (phase, row) order -- and
 * seg_row tells which of the chunk's rows a segment belongs to.  Needs the columns of every row in ascending order.
 * max_rows caps the rows of a chunk (their sums are accumulated in LDS on the device). */
typedef struct { uint32_t cnt; int32_t col; } hubkey_t;
static int cmp_hub(const void *a, const void *b)
{
    const hubkey_t *x = (const hubkey_t *)a, *y = (const hubkey_t *)b;
    if (x->cnt != y->cnt) return x->cnt > y->cnt ? -1 : 1;      /* non-zeros descending */
    return x->col < y->col ? -1 : x->col > y->col;              /* ties by column        */
}

int orc_cvr64_build_ex(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols, const void *vals, int is_f32,
                       int S, int64_t thr, int use_dict, int phases, int64_t max_rows, orc_cvr64 *c)
{
    return orc_cvr64_build_hub(nrows, ncols, rp, cols, vals, is_f32, S, thr, use_dict, phases, max_rows, 0, c);
}

/* hub_max > 0: hub table -- the (at most hub_max) columns with the most non-zeros (at least 2), ranked by non-zeros descending
 * and column ascending; a slot of such a column holds bit 30 and the column's rank instead of the column index */
int orc_cvr64_build_hub(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols, const void *vals, int is_f32,
                        int S, int64_t thr, int use_dict, int phases, int64_t max_rows, int64_t hub_max, orc_cvr64 *c)
{
    return orc_cvr64_build_all(nrows, ncols, rp, cols, vals, is_f32, S, thr, use_dict, phases, max_rows, hub_max, 0, c);
}

/* narrow != 0: narrow chunks -- the column part of a group holds 16-bit offsets from the chunk's smallest column (bit 15 = end
 * of segment, 0x7fff = the pad column); needs every chunk to span fewer than 32 767 columns (-8 otherwise), no dictionary,
 * no phases, no hub table */
int orc_cvr64_build_all(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols, const void *vals, int is_f32,
                        int S, int64_t thr, int use_dict, int phases, int64_t max_rows, int64_t hub_max, int narrow, orc_cvr64 *c)
{
    return orc_cvr64_build_full(nrows, ncols, rp, cols, vals, is_f32, S, thr, use_dict, phases, max_rows, hub_max, 0, narrow, c);
}

/* reorder != 0 (with hub_max > 0): EVERY column index of the image is the column's popularity rank (the device re-orders the
 * whole of x before every SpMV); ranks below the table size also carry bit 30 */
int orc_cvr64_build_full(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols, const void *vals, int is_f32,
                         int S, int64_t thr, int use_dict, int phases, int64_t max_rows, int64_t hub_max, int reorder, int narrow, orc_cvr64 *c)
{
    return orc_cvr64_build_tag(nrows, ncols, rp, cols, vals, is_f32, S, thr, use_dict, phases, max_rows, hub_max, reorder, narrow, 0, 0, c);
}

/* tag16 != 0 (phases > 1): wide row tags -- the chunk's row of every piece stands in a 16-bit tag of its own ([64 lanes][4 x u16]
 * behind the group's column words) instead of above the column index: no limit from the width of the column index.
 * piece_max > 0 (phases > 1): a (row, phase) segment is cut into pieces at the multiples of piece_max elements counted from the
 * chunk's first element, so that no lane holds a piece for longer than that many steps */
int orc_cvr64_build_tag(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols, const void *vals, int is_f32,
                        int S, int64_t thr, int use_dict, int phases, int64_t max_rows, int64_t hub_max, int reorder, int narrow, int tag16, int64_t piece_max, orc_cvr64 *c)
{
    memset(c, 0, sizeof(*c));
    if (tag16 && phases <= 1) tag16 = 0;
    c->tag16 = tag16 != 0;
    if (narrow && (use_dict || phases > 1 || hub_max > 0)) return -8;
    c->narrow = narrow != 0;
    int32_t *hub_index = NULL;
    if (hub_max > 0 && phases <= 1 && ncols > 0 && nrows > 0) {
        hubkey_t *k = (hubkey_t *)calloc((size_t)ncols, sizeof(hubkey_t));
        for (int64_t j = 0; j < ncols; j++) k[j].col = (int32_t)j;
        const int64_t nnz_all = rp[nrows] - rp[0], stride = nnz_all > (1 << 23) ? (nnz_all + (1 << 23) - 1) >> 23 : 1;
        for (int64_t j = rp[0]; j < rp[nrows]; j += stride) k[cols[j]].cnt++;         /* every stride-th non-zero, as the product counts */
        qsort(k, (size_t)ncols, sizeof(hubkey_t), cmp_hub);
        int64_t H = 0;
        while (H < hub_max && H < ncols && k[H].cnt >= 2) H++;
        c->hub_n = (int)H;
        const int64_t keep = reorder && H > 0 ? ncols : H;
        c->order_n = reorder && H > 0 ? (int)ncols : 0;
        c->hub_cols = (int32_t *)calloc((size_t)keep + 1, sizeof(int32_t));
        hub_index = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
        for (int64_t j = 0; j < ncols; j++) hub_index[j] = -1;
        for (int64_t i = 0; i < keep; i++) { c->hub_cols[i] = k[i].col; hub_index[k[i].col] = (int32_t)i; }
        free(k);
    }
    if (phases < 1) phases = 1;
    c->phases = phases;
    int64_t pw = (ncols + phases - 1) / phases;
    pw = (pw + 15) / 16 * 16;
    if (pw < 16) pw = 16;
    int col_bits = 1;                      /* phases: the last column word of a segment carries the chunk's row above the column index */
    while (((int64_t)1 << col_bits) <= ncols) col_bits++;
    if (c->tag16) col_bits = 31;
    c->col_bits = phases > 1 ? col_bits : 31;
    if (phases > 1 && !c->tag16 && (col_bits >= 31 || max_rows > (((int64_t)1 << (31 - col_bits)) - 1) || max_rows <= 0)) return -7;
    if (c->tag16 && (max_rows <= 0 || max_rows > 65534)) return -7;
    if (S < 4 || S % 4) return -1;
    const int64_t cap = (int64_t)W * S;
    if (thr <= 0) thr = cap / 4;
    if (thr > cap / 2) thr = cap / 2;
    c->nrows = nrows; c->ncols = ncols; c->nnz = nrows ? rp[nrows] - rp[0] : 0; c->S = S; c->is_f32 = is_f32;
    chunk_t *ch;
    c->nchunks = plan(nrows, rp, cap, thr, phases > 1 ? max_rows : 0, &ch, &c->shared, &c->nshared);
    const int64_t NC = c->nchunks;
    const int G = S / 4;
    if (use_dict) {
        c->ndict = build_dict(vals, is_f32, nrows ? rp[0] : 0, nrows ? rp[nrows] : 0, c->dict);
        if (c->ndict < 0) { free(ch); return -5; }
    }
    const size_t gb = (c->ndict ? 1280 : c->narrow ? (is_f32 ? 1536 : 2560) : is_f32 ? 2048 : 3072) + (c->tag16 ? 512 : 0);
    const size_t cbytes = (c->narrow ? 512 : 1024) + (c->tag16 ? 512 : 0);          /* what stands in front of a group's values: column part (+ wide row tags) */
    c->image_bytes = (int64_t)((size_t)NC * G * gb);
    if (c->narrow) c->cbase = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
    c->image = (uint8_t *)calloc((size_t)c->image_bytes + 16, 1);
    c->desc = (uint32_t *)calloc((size_t)NC * 4 + 4, sizeof(uint32_t));
    c->target = (uint8_t *)calloc((size_t)NC * W + 1, 1);
    c->nz_begin = (int64_t *)calloc((size_t)NC + 1, sizeof(int64_t));
    c->pad_cnt = (int64_t *)calloc((size_t)NC + 1, sizeof(int64_t));
    int64_t *seg_pos = NULL, *seg_cnt = NULL, *seg_r = NULL, segcap = 0;
    int64_t nsegtot = 0, srcap = 0;
    if (phases > 1) {
        c->seg_off = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
        c->nrows_in = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
    }
    int rc = 0;
    for (int64_t k = 0; k < NC && !rc; k++) {
        const chunk_t *q = &ch[k];
        const int64_t b = q->nzb, e = k + 1 < NC ? ch[k + 1].nzb : (nrows ? rp[nrows] : 0);
        c->nz_begin[k] = b; c->pad_cnt[k] = q->pad;
        c->desc[4 * k + 0] = (uint32_t)q->row_first;
        c->desc[4 * k + 1] = (uint32_t)q->nseg;
        for (int w = 0; w < 2; w++) {      /* destination of the first and of the last segment (phases: of the first and last ROW) */
            const int64_t s = w ? (phases > 1 ? q->nrows_in : q->nseg) - 1 : 0;
            uint32_t d;
            if (s >= q->nrows_in) d = (uint32_t)nrows;                                        /* pad -> dump  */
            else if (s == 0 && q->head) d = (uint32_t)(nrows + 1 + 2 * k);                    /* carry_head   */
            else if (s == q->nrows_in - 1 && q->tail) d = (uint32_t)(nrows + 1 + 2 * k + 1);  /* carry_tail   */
            else d = (uint32_t)(q->row_first + s);
            c->desc[4 * k + 2 + w] = d;
        }
        const int64_t need = phases > 1 ? cap + 1 : q->nseg;       /* a segment holds at least one slot */
        if (need > segcap) {
            segcap = need * 2;
            seg_pos = (int64_t *)realloc(seg_pos, sizeof(int64_t) * (size_t)segcap);
            seg_cnt = (int64_t *)realloc(seg_cnt, sizeof(int64_t) * (size_t)segcap);
            seg_r = (int64_t *)realloc(seg_r, sizeof(int64_t) * (size_t)segcap);
        }
        /* the chunk's segment list: (first CSR element or -1 for pad slots, slot count[, row of the chunk]) */
        int64_t n = 0, tot = 0;
        if (phases == 1) {
            for (int64_t r = q->row_first; r < q->row_first + q->nrows_in; r++) {
                int64_t a = rp[r] > b ? rp[r] : b, z = rp[r + 1] < e ? rp[r + 1] : e;
                if (z > a) { seg_pos[n] = a; seg_cnt[n] = z - a; } else { seg_pos[n] = -1; seg_cnt[n] = 1; }
                tot += seg_cnt[n]; n++;
            }
        } else {
            for (int p = 0; p < phases; p++)
                for (int64_t r = q->row_first; r < q->row_first + q->nrows_in; r++) {
                    int64_t a = rp[r] > b ? rp[r] : b, z = rp[r + 1] < e ? rp[r + 1] : e;
                    if (z <= a) {                      /* empty row: its pad slot goes with phase 0 */
                        if (p == 0) { seg_pos[n] = -1; seg_cnt[n] = 1; seg_r[n] = r - q->row_first; tot++; n++; }
                        continue;
                    }
                    int64_t lo = a, hi;
                    while (lo < z && cols[lo] < (int64_t)p * pw) lo++;
                    hi = lo;
                    while (hi < z && cols[hi] < (int64_t)(p + 1) * pw) hi++;
                    for (int64_t j = a + 1; j < z; j++) if (cols[j] < cols[j - 1]) rc = -6;      /* unsorted row */
                    for (int64_t u = lo; u < hi;) {
                        int64_t v = hi;
                        if (piece_max > 0) { const int64_t nxt = ((u - b) / piece_max + 1) * piece_max + b; if (nxt < v) v = nxt; }
                        seg_pos[n] = u; seg_cnt[n] = v - u; seg_r[n] = r - q->row_first; tot += v - u; n++;
                        u = v;
                    }
                }
        }
        if (q->pad > 0) { seg_pos[n] = -1; seg_cnt[n] = q->pad; seg_r[n] = q->nrows_in; tot += q->pad; n++; }
        if ((phases == 1 && n != q->nseg) || tot != cap) { fprintf(stderr, "cvr64 mirror: chunk %lld holds %lld slots\n", (long long)k, (long long)tot); rc = -2; break; }
        if (phases > 1) {
            c->desc[4 * k + 1] = (uint32_t)n;
            c->seg_off[k] = (uint32_t)nsegtot; c->seg_off[k + 1] = (uint32_t)(nsegtot + n);
            c->nrows_in[k] = (uint32_t)q->nrows_in;
            if (nsegtot + n > srcap) { srcap = (nsegtot + n) * 2; c->seg_row = (uint16_t *)realloc(c->seg_row, sizeof(uint16_t) * (size_t)srcap); }
            for (int64_t i = 0; i < n; i++) c->seg_row[nsegtot + i] = (uint16_t)seg_r[i];
            nsegtot += n;
        }

        if (c->narrow) {                  /* the chunk's smallest column; its span must fit 15 bits minus the pad code */
            int64_t lo = INT64_MAX, hi = -1;
            for (int64_t j = b; j < e; j++) { if (cols[j] < lo) lo = cols[j]; if (cols[j] > hi) hi = cols[j]; }
            c->cbase[k] = hi >= 0 ? (uint32_t)lo : 0u;
            if (hi >= 0 && hi - lo >= 0x7fff) { rc = -8; break; }
        }
        int64_t pos[W], cnt[W], fed = 0;
        uint32_t tag[W];
        for (int l = 0; l < W; l++) { pos[l] = -1; cnt[l] = 0; tag[l] = 0; c->target[k * W + l] = (uint8_t)l; }
        for (int i = 0; i < S && !rc; i++) {
            const int64_t ave = S - i;      /* == sum(cnt)/64 */
            for (int l = 0; l < W; l++) {   /* empty lanes in lane order (spmv.cpp:814-816) */
                if (cnt[l] != 0) continue;
                if (fed < n) { pos[l] = seg_pos[fed]; cnt[l] = seg_cnt[fed]; tag[l] = phases > 1 ? (c->tag16 ? (uint32_t)seg_r[fed] : (uint32_t)seg_r[fed] << col_bits) : 0; fed++; }     /* spmv.cpp:821-868 */
                else {
                    int v = 0;
                    while (v < W && cnt[v] <= ave) v++;      /* first over-full lane (spmv.cpp:876-879) */
                    if (v == W) { fprintf(stderr, "cvr64 mirror: no victim\n"); rc = -3; break; }
                    pos[l] = pos[v]; cnt[l] = ave; tag[l] = tag[v];   /* the stealer takes the FIRST ave (spmv.cpp:927-931); phases: the stolen piece keeps the row of the victim's segment */
                    if (pos[v] >= 0) pos[v] += ave;
                    cnt[v] -= ave;
                    c->target[k * W + l] = (uint8_t)v;
                }
            }
            const int g = i / 4, j = i % 4;
            uint8_t *grp = c->image + ((size_t)k * G + g) * gb;
            for (int l = 0; l < W; l++) {
                uint32_t col = (uint32_t)ncols; double v = 0;
                if (pos[l] >= 0) {
                    col = (uint32_t)cols[pos[l]];
                    if (hub_index && hub_index[col] >= 0) col = (hub_index[col] < c->hub_n ? 0x40000000u : 0u) | (uint32_t)hub_index[col];
                    v = is_f32 ? (double)((const float *)vals)[pos[l]] : ((const double *)vals)[pos[l]];
                    pos[l]++;
                }
                if (cnt[l] == 1) col |= 0x80000000u | (c->tag16 ? 0u : tag[l]);
                if (c->tag16) ((uint16_t *)(grp + 1024))[l * 4 + j] = (uint16_t)(cnt[l] == 1 ? tag[l] : 0u);
                if (c->narrow) {
                    const uint32_t cc = col & 0x7fffffffu;
                    ((uint16_t *)grp)[l * 4 + j] = (uint16_t)((cc == (uint32_t)ncols ? 0x7fffu : cc - c->cbase[k]) | (col >> 31 << 15));
                } else
                ((uint32_t *)grp)[l * 4 + j] = col;
                if (c->ndict) {                               /* one code byte per slot: position in the sorted dictionary */
                    uint64_t b;
                    if (is_f32) { float f = (float)v; uint32_t u; memcpy(&u, &f, 4); b = u; } else memcpy(&b, &v, 8);
                    int code = 0;
                    while (code < c->ndict && c->dict[code] != b) code++;
                    (grp + cbytes)[l * 4 + j] = (uint8_t)code;
                }
                else if (is_f32) ((float *)(grp + cbytes))[l * 4 + j] = (float)v;
                else ((double *)(grp + cbytes + (j / 2) * 1024))[l * 2 + j % 2] = v;
                cnt[l]--;
            }
        }
        for (int l = 0; l < W && !rc; l++)
            if (cnt[l] != 0) { fprintf(stderr, "cvr64 mirror: lane not drained\n"); rc = -4; }
    }
    if (NC) c->nz_begin[NC] = nrows ? rp[nrows] : 0;
    free(seg_pos); free(seg_cnt); free(seg_r); free(ch); free(hub_index);
    return rc;
}

/* Interleaved chunks (cvr_options.interleave; cvr_amd/csrc/cvr_ilv.hip): the chunk plan of an image with column phases (rows capped
 * at max_rows, their sums accumulated in LDS), but the chunk's non-zeros are dealt to the lanes in COLUMN order -- element e of the
 * chunk's list sorted by (column, position) stands at step e / 64, lane e % 64 -- and every slot is a piece of its own: end flag and
 * row in every column word (or 16-bit tag).  Slots behind the chunk's non-zeros (the tail padding and the pad slots the planner counts
 * for empty rows) hold the pad column, value 0 and the dump entry.  Written from that definition; orc_cvr64_spmv interprets the image
 * as it does any image with column phases (c->phases = 2 marks it). */
typedef struct { int32_t col; int64_t pos; } ilvkey_t;
static int cmp_ilv(const void *a, const void *b)
{
    const ilvkey_t *x = (const ilvkey_t *)a, *y = (const ilvkey_t *)b;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    return x->pos < y->pos ? -1 : x->pos > y->pos;
}

int orc_cvr64_build_ilv(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols, const void *vals, int is_f32,
                        int S, int64_t thr, int use_dict, int64_t max_rows, int tag16, orc_cvr64 *c)
{
    memset(c, 0, sizeof(*c));
    c->tag16 = tag16 != 0;
    c->phases = 2;
    int col_bits = 1;
    while (((int64_t)1 << col_bits) <= ncols) col_bits++;
    if (c->tag16) col_bits = 31;
    c->col_bits = col_bits;
    c->ilv = 1;          /* every slot is a piece of its own: no end flag; without tags the column word's bits [col_bits, 32) hold the row */
    if (!c->tag16 && (col_bits >= 31 || max_rows > (((int64_t)1 << (32 - col_bits)) - 1) || max_rows <= 0)) return -7;
    if (c->tag16 && (max_rows <= 0 || max_rows > 65534)) return -7;
    if (S < 4 || S % 4) return -1;
    const int64_t cap = (int64_t)W * S;
    if (thr <= 0) thr = cap / 4;
    if (thr > cap / 2) thr = cap / 2;
    c->nrows = nrows; c->ncols = ncols; c->nnz = nrows ? rp[nrows] - rp[0] : 0; c->S = S; c->is_f32 = is_f32;
    chunk_t *ch;
    c->nchunks = plan(nrows, rp, cap, thr, max_rows, &ch, &c->shared, &c->nshared);
    const int64_t NC = c->nchunks;
    const int G = S / 4;
    if (use_dict) {
        c->ndict = build_dict(vals, is_f32, nrows ? rp[0] : 0, nrows ? rp[nrows] : 0, c->dict);
        if (c->ndict < 0) { free(ch); return -5; }
    }
    const size_t gb = (c->ndict ? 1280 : is_f32 ? 2048 : 3072) + (c->tag16 ? 512 : 0);
    const size_t cbytes = 1024 + (c->tag16 ? 512 : 0);
    c->image_bytes = (int64_t)((size_t)NC * G * gb);
    c->image = (uint8_t *)calloc((size_t)c->image_bytes + 16, 1);
    c->desc = (uint32_t *)calloc((size_t)NC * 4 + 4, sizeof(uint32_t));
    c->target = (uint8_t *)calloc((size_t)NC * W + 1, 1);
    c->nz_begin = (int64_t *)calloc((size_t)NC + 1, sizeof(int64_t));
    c->pad_cnt = (int64_t *)calloc((size_t)NC + 1, sizeof(int64_t));
    c->seg_off = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
    c->nrows_in = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
    c->seg_row = (uint16_t *)calloc(1, sizeof(uint16_t));
    ilvkey_t *key = (ilvkey_t *)malloc(sizeof(ilvkey_t) * (size_t)(cap + 1));
    int code0 = 0;                          /* the dictionary code of +0.0 */
    while (code0 < c->ndict && c->dict[code0] != 0) code0++;
    int rc = 0;
    for (int64_t k = 0; k < NC && !rc; k++) {
        const chunk_t *q = &ch[k];
        const int64_t b = q->nzb, e = k + 1 < NC ? ch[k + 1].nzb : (nrows ? rp[nrows] : 0), n = e - b;
        c->nz_begin[k] = b; c->pad_cnt[k] = q->pad;
        c->nrows_in[k] = (uint32_t)q->nrows_in;
        c->desc[4 * k + 0] = (uint32_t)q->row_first;
        c->desc[4 * k + 1] = (uint32_t)q->nseg;
        for (int w = 0; w < 2; w++) {      /* destination of the first and of the last ROW */
            const int64_t s = w ? q->nrows_in - 1 : 0;
            uint32_t d;
            if (s == 0 && q->head) d = (uint32_t)(nrows + 1 + 2 * k);
            else if (s == q->nrows_in - 1 && q->tail) d = (uint32_t)(nrows + 1 + 2 * k + 1);
            else d = (uint32_t)(q->row_first + s);
            c->desc[4 * k + 2 + w] = d;
        }
        if (n > cap) { rc = -2; break; }
        for (int64_t i = 0; i < n; i++) { key[i].col = cols[b + i]; key[i].pos = b + i; }
        qsort(key, (size_t)n, sizeof(ilvkey_t), cmp_ilv);
        for (int64_t s = 0; s < cap; s++) {
            const int i = (int)(s / W), l = (int)(s % W), g = i / 4, j = i % 4;
            uint8_t *grp = c->image + ((size_t)k * G + g) * gb;
            uint32_t col = (uint32_t)ncols, row = (uint32_t)q->nrows_in;
            double v = 0;
            if (s < n) {
                const int64_t p = key[s].pos;
                int64_t r = q->row_first;            /* the chunk's row of position p: the last of its rows that starts at or before p */
                while (r + 1 < q->row_first + q->nrows_in && rp[r + 1] <= p) r++;
                col = (uint32_t)key[s].col; row = (uint32_t)(r - q->row_first);
                v = is_f32 ? (double)((const float *)vals)[p] : ((const double *)vals)[p];
            }
            ((uint32_t *)grp)[l * 4 + j] = c->tag16 ? col | 0x80000000u : col | row << col_bits;
            if (c->tag16) ((uint16_t *)(grp + 1024))[l * 4 + j] = (uint16_t)row;
            if (c->ndict) {
                int code = code0;
                if (s < n) {
                    uint64_t bits;
                    if (is_f32) { float f = (float)v; uint32_t u; memcpy(&u, &f, 4); bits = u; } else memcpy(&bits, &v, 8);
                    code = 0;
                    while (code < c->ndict && c->dict[code] != bits) code++;
                }
                (grp + cbytes)[l * 4 + j] = (uint8_t)code;
            }
            else if (is_f32) ((float *)(grp + cbytes))[l * 4 + j] = (float)v;
            else ((double *)(grp + cbytes + (j / 2) * 1024))[l * 2 + j % 2] = v;
        }
    }
    if (NC) c->nz_begin[NC] = nrows ? rp[nrows] : 0;
    free(key); free(ch);
    return rc;
}


/* Gang chunks (cvr_options.gang; cvr_amd/csrc/cvr_ilv.hip: gang_write_kernel, cvr_spmv.hip: spmv_gang_kernel).  Written from the definition in
 * cvr_format.h: the plan is that of interleaved chunks; the `gang` consecutive chunks k0 .. k0 + gang - 1 of a workgroup put their non-zeros into ONE list
 * sorted by (column, position); element e stands in group e / 256 of the gang's stream = the chunks' allocations one behind the other, at step (e / 64) % 4,
 * lane e % 64; its tag = (chunk - k0) * ystage + row inside its chunk (the chunk of a position: the last whose first position is at or before it).  The slots
 * behind the last element of the gang's last group: offset 0 (or the pad column with tags), the dump entry of the gang's first chunk, value 0; groups
 * behind that one are not written (zeros).  gbase[k0 * S/4 + g] = column of element 256 g; ggroups[k0] = groups that hold non-zeros. */
int orc_cvr64_build_gang(int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *cols, const void *vals, int is_f32,
                         int S, int64_t thr, int use_dict, int64_t max_rows, int tag16, int gang, int ystage, orc_cvr64 *c)
{
    memset(c, 0, sizeof(*c));
    if (gang < 2 || gang > 16 || ystage < 1 || max_rows <= 0 || max_rows >= ystage) return -7;
    if ((int64_t)gang * ystage > (tag16 ? 65536 : 32768)) return -7;
    c->tag16 = tag16 != 0;
    c->phases = 2;
    c->col_bits = c->tag16 ? 31 : 17;
    c->ilv = 1;
    c->gang = gang; c->ystage = ystage;
    if (S < 4 || S % 4) return -1;
    const int64_t cap = (int64_t)W * S;
    if (thr <= 0) thr = cap / 4;
    if (thr > cap / 2) thr = cap / 2;
    c->nrows = nrows; c->ncols = ncols; c->nnz = nrows ? rp[nrows] - rp[0] : 0; c->S = S; c->is_f32 = is_f32;
    chunk_t *ch;
    c->nchunks = plan(nrows, rp, cap, thr, max_rows, &ch, &c->shared, &c->nshared);
    const int64_t NC = c->nchunks;
    const int G = S / 4;
    if (use_dict) {
        c->ndict = build_dict(vals, is_f32, nrows ? rp[0] : 0, nrows ? rp[nrows] : 0, c->dict);
        if (c->ndict < 0) { free(ch); return -5; }
    }
    const size_t gb = (c->ndict ? 1280 : is_f32 ? 2048 : 3072) + (c->tag16 ? 512 : 0);
    const size_t cbytes = 1024 + (c->tag16 ? 512 : 0);
    c->image_bytes = (int64_t)((size_t)NC * G * gb);
    c->image = (uint8_t *)calloc((size_t)c->image_bytes + 16, 1);
    c->desc = (uint32_t *)calloc((size_t)NC * 4 + 4, sizeof(uint32_t));
    c->target = (uint8_t *)calloc((size_t)NC * W + 1, 1);
    c->nz_begin = (int64_t *)calloc((size_t)NC + 1, sizeof(int64_t));
    c->pad_cnt = (int64_t *)calloc((size_t)NC + 1, sizeof(int64_t));
    c->seg_off = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
    c->nrows_in = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
    c->seg_row = (uint16_t *)calloc(1, sizeof(uint16_t));
    c->gbase = (uint32_t *)calloc((size_t)NC * G + 1, sizeof(uint32_t));
    c->ggroups = (uint32_t *)calloc((size_t)NC + 1, sizeof(uint32_t));
    ilvkey_t *key = (ilvkey_t *)malloc(sizeof(ilvkey_t) * (size_t)(cap * gang + 1));
    int code0 = 0;
    while (code0 < c->ndict && c->dict[code0] != 0) code0++;
    int rc = 0;
    for (int64_t k = 0; k < NC; k++) {          /* the chunks' tables: those of interleaved chunks */
        const chunk_t *q = &ch[k];
        c->nz_begin[k] = q->nzb; c->pad_cnt[k] = q->pad;
        c->nrows_in[k] = (uint32_t)q->nrows_in;
        c->desc[4 * k + 0] = (uint32_t)q->row_first;
        c->desc[4 * k + 1] = (uint32_t)q->nseg;
        for (int w = 0; w < 2; w++) {
            const int64_t s = w ? q->nrows_in - 1 : 0;
            uint32_t d;
            if (s == 0 && q->head) d = (uint32_t)(nrows + 1 + 2 * k);
            else if (s == q->nrows_in - 1 && q->tail) d = (uint32_t)(nrows + 1 + 2 * k + 1);
            else d = (uint32_t)(q->row_first + s);
            c->desc[4 * k + 2 + w] = d;
        }
        if (q->nrows_in >= ystage) rc = -7;
    }
    if (NC) c->nz_begin[NC] = nrows ? rp[nrows] : 0;
    for (int64_t k0 = 0; k0 < NC && !rc; k0 += gang) {
        const int64_t nc = NC - k0 < gang ? NC - k0 : gang;
        const int64_t b = c->nz_begin[k0], e = c->nz_begin[k0 + nc], n = e - b;
        if (n > cap * nc) { rc = -2; break; }
        for (int64_t i = 0; i < n; i++) { key[i].col = cols[b + i]; key[i].pos = b + i; }
        qsort(key, (size_t)n, sizeof(ilvkey_t), cmp_ilv);
        const int64_t GGn = (n + 255) / 256;
        c->ggroups[k0] = (uint32_t)GGn;
        const uint32_t dump = (uint32_t)ch[k0].nrows_in;          /* the first chunk's dump entry */
        for (int64_t s = 0; s < GGn * 256; s++) {
            const int64_t g = s / 256;
            const int j = (int)((s / W) % 4), l = (int)(s % W);
            uint8_t *grp = c->image + ((size_t)k0 * G + (size_t)g) * gb;
            const uint32_t bcol = (uint32_t)key[g * 256].col;
            if (s % 256 == 0 && !c->tag16) c->gbase[(size_t)k0 * G + (size_t)g] = bcol;
            uint32_t col = c->tag16 ? (uint32_t)ncols : bcol, tag = dump;
            double v = 0;
            if (s < n) {
                const int64_t p = key[s].pos;
                int64_t kc = k0;
                while (kc + 1 < k0 + nc && c->nz_begin[kc + 1] <= p) kc++;
                const chunk_t *q = &ch[kc];
                int64_t r = q->row_first;
                while (r + 1 < q->row_first + q->nrows_in && rp[r + 1] <= p) r++;
                col = (uint32_t)key[s].col; tag = (uint32_t)((kc - k0) * ystage + (r - q->row_first));
                v = is_f32 ? (double)((const float *)vals)[p] : ((const double *)vals)[p];
            }
            if (c->tag16) { ((uint32_t *)grp)[l * 4 + j] = col | 0x80000000u; ((uint16_t *)(grp + 1024))[l * 4 + j] = (uint16_t)tag; }
            else {
                const uint32_t off = col - bcol;
                if (off >> 17) rc = -9;
                ((uint32_t *)grp)[l * 4 + j] = (off & 0x1ffffu) | tag << 17;
            }
            if (c->ndict) {
                int code = code0;
                if (s < n) {
                    uint64_t bits;
                    if (is_f32) { float f = (float)v; uint32_t u; memcpy(&u, &f, 4); bits = u; } else memcpy(&bits, &v, 8);
                    code = 0;
                    while (code < c->ndict && c->dict[code] != bits) code++;
                }
                (grp + cbytes)[l * 4 + j] = (uint8_t)code;
            }
            else if (is_f32) ((float *)(grp + cbytes))[l * 4 + j] = (float)v;
            else ((double *)(grp + cbytes + (j / 2) * 1024))[l * 2 + j % 2] = v;
        }
    }
    free(key); free(ch);
    return rc;
}

void orc_cvr64_free(orc_cvr64 *c)
{
    free(c->image); free(c->desc); free(c->target); free(c->shared); free(c->nz_begin); free(c->pad_cnt);
    free(c->seg_off); free(c->seg_row); free(c->nrows_in); free(c->hub_cols); free(c->cbase); free(c->gbase); free(c->ggroups);
    memset(c, 0, sizeof(*c));
}

/* Interpret a CVR64 image by the format's write-back rules (the HIP kernel is cvr_amd/csrc/cvr_spmv.hip):
 *   a lane's segment ends while segments remain   -> y_ext[dest(cur)] = acc; take segment fed+rank
 *   ... ends as the last segments are handed out  -> y_ext[dest(cur)] = acc; the lane turns stealer
 *   ... ends after the last segment was handed out-> slot[lane] = acc;      the lane turns stealer
 *   end of chunk: slot[target[lane]] += acc of lanes that stole; owners store their slot
 *   column phases: none of the above -- the last column word of EVERY piece (a segment, or what a lane stole of one) carries
 *   the chunk's row it belongs to, and the piece's sum is added to that row's accumulator when the piece ends (step by step,
 *   lanes in order), the accumulators are written out at the end of the chunk
 *   fix-up: y[row] = carry_tail(c0) + sum of carry_head(c) for c0 < c <= c1, in chunk order */
void orc_cvr64_spmv(const orc_cvr64 *c, const void *xv, void *yv)
{
    const int S = c->S, G = S / 4;
    const int64_t NC = c->nchunks, nrows = c->nrows;
    const size_t gb = (c->ndict ? 1280 : c->narrow ? (c->is_f32 ? 1536 : 2560) : c->is_f32 ? 2048 : 3072) + (c->tag16 ? 512 : 0);
    const size_t cbytes = (c->narrow ? 512 : 1024) + (c->tag16 ? 512 : 0);
    const size_t next = (size_t)(nrows + 1 + 2 * NC);
    const int ph = c->phases > 1;          /* column phases: a segment's sum is ADDED to its row's accumulator (LDS on the device) */
    double *yext = (double *)calloc(next + 1, sizeof(double));
    double *yloc = (double *)calloc((size_t)W * S + 2, sizeof(double));
    /* gang chunks: the gang's list is walked group by group, step by step, lane by lane -- the order the token of spmv_gang_kernel enforces whatever its
     * wavefronts' timing --: every slot's rounded product is added to the accumulator its tag names; then every chunk writes its rows */
    if (c->gang) {
        double *ya = (double *)calloc((size_t)c->gang * c->ystage + 1, sizeof(double));
        for (int64_t k0 = 0; k0 < NC; k0 += c->gang) {
            const int64_t nc = NC - k0 < c->gang ? NC - k0 : c->gang;
            for (int64_t i = 0; i < (int64_t)c->gang * c->ystage; i++) ya[i] = 0;
            for (uint32_t g = 0; g < c->ggroups[k0]; g++) {
                const uint8_t *grp = c->image + ((size_t)k0 * G + g) * gb;
                const uint32_t bcol = c->gbase[(size_t)k0 * G + g];
                for (int j = 0; j < 4; j++)
                    for (int l = 0; l < W; l++) {
                        const uint32_t cw = ((const uint32_t *)grp)[l * 4 + j];
                        const uint32_t col = c->tag16 ? cw & 0x7fffffffu : bcol + (cw & 0x1ffffu);
                        const uint32_t tag = c->tag16 ? ((const uint16_t *)(grp + 1024))[l * 4 + j] : cw >> 17;
                        if (c->is_f32) {
                            float v;
                            if (c->ndict) { const uint32_t u = (uint32_t)c->dict[(grp + cbytes)[l * 4 + j]]; memcpy(&v, &u, 4); }
                            else v = ((const float *)(grp + cbytes))[l * 4 + j];
                            ya[tag] = (double)((float)ya[tag] + fmaf(v, ((const float *)xv)[col], 0.0f));
                        } else {
                            double v;
                            if (c->ndict) memcpy(&v, &c->dict[(grp + cbytes)[l * 4 + j]], 8);
                            else v = ((const double *)(grp + cbytes + (j / 2) * 1024))[l * 2 + j % 2];
                            ya[tag] += fma(v, ((const double *)xv)[col], 0.0);
                        }
                    }
            }
            for (int64_t kc = k0; kc < k0 + nc; kc++) {
                const uint32_t row_first = c->desc[4 * kc], hd = c->desc[4 * kc + 2], ld = c->desc[4 * kc + 3], nri = c->nrows_in[kc];
                for (uint32_t i = 0; i < nri; i++) yext[i == 0 ? hd : i == nri - 1 ? ld : row_first + i] = ya[(size_t)(kc - k0) * c->ystage + i];
            }
        }
        free(ya);
    }
    for (int64_t k = 0; k < NC && !c->gang; k++) {
        const uint32_t row_first = c->desc[4 * k], n = c->desc[4 * k + 1], hd = c->desc[4 * k + 2], ld = c->desc[4 * k + 3];
        const uint32_t nri = ph ? c->nrows_in[k] : 0;
        const uint32_t cmask = ph && !c->tag16 ? (1u << c->col_bits) - 1u : 0x7fffffffu;
        const uint32_t nd = ph ? nri : n;      /* DEST is indexed by the segment (implicit rows) or by the chunk's row (phases) */
#define DEST(q) ((q) == 0 ? hd : (q) == nd - 1 ? ld : row_first + (q))
#define ADDTO(dst, v) do { if (c->is_f32) (dst) = (double)((float)(dst) + (float)(v)); else (dst) += (v); } while (0)
        double acc[W], slot[W];
        uint32_t cur[W], rowtag[W];
        int feeding[W], own[W];
        uint32_t fed = n < W ? n : W;
        for (int l = 0; l < W; l++) { acc[l] = 0; slot[l] = 0; feeding[l] = (uint32_t)l < fed; own[l] = 0; cur[l] = (uint32_t)l; }
        if (ph) for (uint32_t i = 0; i <= nri; i++) yloc[i] = 0;
        int tail = fed == n;
        for (int i = 0; i < S; i++) {
            const uint8_t *grp = c->image + ((size_t)k * G + i / 4) * gb;
            const int j = i % 4;
            int flagged[W];
            for (int l = 0; l < W; l++) {
                uint32_t cw;
                if (c->narrow) {
                    const uint32_t h = ((const uint16_t *)grp)[l * 4 + j], off = h & 0x7fffu;
                    cw = (off == 0x7fffu ? (uint32_t)c->ncols : c->cbase[k] + off) | ((h >> 15) << 31);
                } else cw = ((const uint32_t *)grp)[l * 4 + j];
                uint32_t col = cw & cmask;
                if (c->hub_n && (col & 0x40000000u)) col = (uint32_t)c->hub_cols[col & 0x3fffffffu];     /* hub slot: rank -> column */
                else if (c->order_n && col < (uint32_t)c->order_n) col = (uint32_t)c->hub_cols[col];      /* re-ordered x: rank -> column (the pad column stays) */
                flagged[l] = c->ilv ? 1 : cw >> 31;
                rowtag[l] = !ph ? 0 : c->tag16 ? ((const uint16_t *)(grp + 1024))[l * 4 + j] : c->ilv ? cw >> c->col_bits : (cw & 0x7fffffffu) >> c->col_bits;
                if (c->is_f32) {
                    float v;
                    if (c->ndict) { const uint32_t u = (uint32_t)c->dict[(grp + cbytes)[l * 4 + j]]; memcpy(&v, &u, 4); }
                    else v = ((const float *)(grp + cbytes))[l * 4 + j];
                    acc[l] = (double)fmaf(v, ((const float *)xv)[col], (float)acc[l]);
                } else {
                    double v;
                    if (c->ndict) memcpy(&v, &c->dict[(grp + cbytes)[l * 4 + j]], 8);
                    else v = ((const double *)(grp + cbytes + (j / 2) * 1024))[l * 2 + j % 2];
                    acc[l] = fma(v, ((const double *)xv)[col], acc[l]);
                }
            }
            if (ph) {          /* column phases: every piece -- fed or stolen -- carries its row and adds its sum to that row's accumulator when it ends */
                for (int l = 0; l < W; l++)
                    if (flagged[l]) { ADDTO(yloc[rowtag[l]], acc[l]); acc[l] = 0; }
            } else if (!tail) {
                uint32_t rank = 0;
                for (int l = 0; l < W; l++) {
                    if (!flagged[l]) continue;
                    yext[DEST(cur[l])] = acc[l];
                    acc[l] = 0;
                    if (fed + rank < n) cur[l] = fed + rank; else feeding[l] = 0;
                    rank++;
                }
                fed += rank;
                if (fed >= n) { fed = n; tail = 1; }
            } else {
                for (int l = 0; l < W; l++)
                    if (flagged[l] && feeding[l]) { slot[l] = acc[l]; acc[l] = 0; feeding[l] = 0; own[l] = 1; }
            }
        }
        for (int l = 0; l < W && !ph; l++) {
            const int t = c->target[k * W + l];
            if (t == l) continue;
            ADDTO(slot[t], acc[l]);
        }
        for (int l = 0; l < W && !ph; l++)
            if (own[l]) yext[DEST(cur[l])] = slot[l];
        if (ph) for (uint32_t i = 0; i < nri; i++) yext[DEST(i)] = yloc[i];
#undef DEST
#undef ADDTO
    }
    free(yloc);
    for (int64_t s = 0; s < c->nshared; s++) {
        const int64_t row = c->shared[3 * s], c0 = c->shared[3 * s + 1], c1 = c->shared[3 * s + 2];
        double v = 0;
        for (int64_t k = c0 + 1; k <= c1; k++) {
            if (c->is_f32) v = (double)((float)v + (float)yext[nrows + 1 + 2 * k]);
            else v += yext[nrows + 1 + 2 * k];
        }
        yext[row] = c->is_f32 ? (double)((float)yext[nrows + 1 + 2 * c0 + 1] + (float)v) : yext[nrows + 1 + 2 * c0 + 1] + v;
    }
    if (c->is_f32) for (int64_t r = 0; r < nrows; r++) ((float *)yv)[r] = (float)yext[r];
    else memcpy(yv, yext, sizeof(double) * (size_t)nrows);
    free(yext);
}

/* oracle/cvr_oracle.h -- TEST INFRASTRUCTURE.  NOT part of the product.
 *
 * Plain-C CPU restatement of the reference's hot path (puckbee/CVR, /root/reference/spmv.cpp):
 *   orc_read_matrix   <- readMatrix            spmv.cpp:311-535   (Matrix-Market -> padded 1-based CSR)
 *   orc_csr_spmv      <- the CSR self-check    spmv.cpp:1843-1850 (THE parity oracle)
 *   orc_cvr8_convert  <- pre_processing        spmv.cpp:565-1014  (CSR -> 8-lane CVR, reference layout)
 *   orc_cvr8_spmv     <- spmv_compute_kernel   spmv.cpp:1016-1667 (y = A x over the 8-lane layout,
 *                                                                  bugs K1/K2 of SURVEY App. B fixed)
 * plus (cvr64_mirror.c) a CPU mirror of THIS repo's 64-lane device format, used only to check the
 * HIP converter bit-for-bit and to interpret a CVR64 image without a GPU.
 *
 * Pinned by tests/golden/NAME.npz, which hold the outputs of the unmodified reference run in the
 * build container (oracle/gen_fixtures.py).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 */
#ifndef CVR_ORACLE_H
#define CVR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference-compatible CSR (1-based; numRows+2 row pointers; nnz padded to a multiple of 16) ---- */
typedef struct {
    int     nItems;     /* padded nnz                      (spmv.cpp:390, 457, 490)            */
    int     nItemsRaw;  /* entries read incl. mirrors      (spmv.cpp:455)                      */
    int     numRows;    /* header value                    (spmv.cpp:386)                      */
    int     numCols;
    double *val;        /* [nItems]  fp32-rounded          (spmv.cpp:65, 432-433, 512)         */
    int    *cols;       /* [nItems]  1-based               (spmv.cpp:437-438 commented out)    */
    int    *rowptr;     /* [numRows+2]; tail = nItems-1    (spmv.cpp:499-526)                  */
} orc_csr;

int  orc_read_matrix(const char *path, orc_csr *out);  /* 0 ok, <0 error (reference: exit(1)) */
void orc_free_csr(orc_csr *m);

/* y[i] = sum_j val[j]*x[cols[j]], j ascending, rows 0..numRows-1 (spmv.cpp:1843-1850).
 * x must hold numCols+1 entries (1-based columns). */
void orc_csr_spmv(int numRows, const int *rowptr, const int *cols, const double *val,
                  const double *x, double *y);
/* same loop over any 0-based CSR with 64-bit row pointers; also returns sum_j |a_ij x_j| per row
 * (the scale of the stated tolerance) when absy != NULL */
void orc_csr_spmv64(int64_t nrows, const int64_t *rowptr, const int32_t *cols, const double *val,
                    const double *x, double *y, double *absy);
void orc_csr_spmv64_f32(int64_t nrows, const int64_t *rowptr, const int32_t *cols, const float *val,
                        const float *x, double *y, double *absy);

/* seeded non-constant x: splitmix64(0xC0FFEE, j) -> uniform [-1,1)   */
double orc_x_rand(uint64_t j);

/* bench.py helper: row-major `pattern general` .mtx of a 0-based CSR pattern (input for oracle/_ref) */
int orc_write_mtx_pattern(const char *path, int64_t nrows, int64_t ncols, const int64_t *rp, const int32_t *ci);

/* ---- 8-lane CVR in the reference's own layout ---- */
typedef struct {
    int     T;          /* chunks = reference threads                                           */
    int     nItems, numRows;
    double *vals;       /* [nItems]  step-major [step][lane] per chunk       (A.3)              */
    int    *cols;       /* [nItems]                                                              */
    int    *record;     /* [2*(numRows+240+32T)] (pos,wb) pairs              (A.5, A.7, A.8)    */
    int64_t record_len;
    int    *split;      /* [2T]  {ncsr_start, ncsr}                          (A.5)              */
    int    *final2;     /* [16T] tail rows                                   (A.5, A.7)         */
    int    *nnz_rows;   /* [4T]  {s_t, e_t, first_row, last_row}             (A.2)              */
} orc_cvr8;

#define ORC_RECORD_SENTINEL (-0x7f7f7f7f)
int  orc_cvr8_convert(const orc_csr *m, int T, orc_cvr8 *out);   /* <0 if nItems < 16*T */
void orc_cvr8_free(orc_cvr8 *c);
/* y (numRows+2 entries, zeroed inside) = A x; nthreads OpenMP threads over the T chunks */
void orc_cvr8_spmv(const orc_cvr8 *c, const double *x, double *y, int nthreads);

/* ---- CPU mirror of the repo's 64-lane device format (cvr64_mirror.c) ---- */
typedef struct {
    int64_t nrows, ncols, nnz;
    int     S;               /* steps per chunk (multiple of 4)                                  */
    int     is_f32;
    int64_t nchunks;
    int64_t nshared;         /* rows cut over several chunks                                     */
    int64_t image_bytes;     /* nchunks * S/4 * (3072 | 2048)                                    */
    uint8_t  *image;         /* per group: [64][4] u32 column words (bit 31 = segment end), then */
                             /* f64: [2][64][2] values; f32: [64][4] values                      */
    uint32_t *desc;          /* [nchunks][4] = {row_first, nseg, head_dest, last_dest}           */
    uint8_t  *target;        /* [nchunks][64] lane stolen from (itself if none)                  */
    int64_t  *shared;        /* [nshared][3] = {row, first chunk, last chunk}                    */
    int64_t  *nz_begin;      /* [nchunks+1] plan: first CSR element of each chunk                */
    int64_t  *pad_cnt;       /* [nchunks]   plan: slots of the trailing pad segment              */
    int       ndict;         /* 0, or entries of the value dictionary: the group then holds       */
                             /* [64][4] u8 codes (256 B) after the column words instead of values */
    uint64_t  dict[256];     /* distinct value bit patterns + 0, sorted                           */
    int       phases;        /* 1, or the number of column phases (orc_cvr64_build_ex): desc[k].nseg then counts the   */
                             /* (row, phase) segments + the pad segment, head/last_dest belong to the first / last ROW */
    uint32_t *seg_off;       /* phases > 1: [nchunks+1] first entry of chunk k in seg_row                               */
    uint16_t *seg_row;       /* phases > 1: the chunk's row (0 .. nrows_in-1; nrows_in = dump) of every segment        */
    uint32_t *nrows_in;      /* phases > 1: [nchunks] rows with a segment in the chunk                                  */
    int       col_bits;      /* phases > 1: the last column word of a segment holds its seg_row in bits [col_bits, 31)   */
    int       hub_n;         /* hub table entries (0 = none): a column word with bit 30 holds the rank of a hub column      */
    int32_t  *hub_cols;      /* [hub_n] the hub columns by non-zeros descending, ties by column                             */
    int       order_n;       /* 0, or ncols: every column index of the image is a popularity rank (hub_cols then lists all columns) */
    int       narrow;        /* 1: narrow chunks -- groups hold [64][4] u16 column offsets from cbase[k] (512 B) before the values */
    uint32_t *cbase;         /* narrow: [nchunks] smallest column of the chunk                                              */
    int       tag16;         /* phases > 1: the rows of the pieces stand in [64][4] u16 tags (512 B) behind the column words (col_bits = 31) */
    int       ilv;           /* interleaved chunks (orc_cvr64_build_ilv): every slot ends a piece; without tags its column word holds the row in bits [col_bits, 32) */
    int       gang;          /* gang chunks (orc_cvr64_build_gang): `gang` consecutive chunks share one sorted list; 0 = none                                      */
    int       ystage;        /* gang chunks: accumulators per chunk -- a slot's tag = chunk inside the gang * ystage + row inside the chunk                        */
    uint32_t *gbase;         /* gang chunks: [nchunks * S/4] the first column of every group of every gang (gang b's groups from (b * gang) * S/4 on; zeros behind its last, and with 16-bit tags) */
    uint32_t *ggroups;       /* gang chunks: [nchunks] groups of the gang that hold non-zeros, at the gang's first chunk (0 at the others)                         */
} orc_cvr64;

int  orc_cvr64_build(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols,
                     const void *vals, int is_f32, int S, int64_t split_threshold, orc_cvr64 *out);
/* the same with a value dictionary (use_dict != 0): -5 when the matrix has more than 256 distinct values */
int  orc_cvr64_build_dict(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols,
                          const void *vals, int is_f32, int S, int64_t split_threshold, int use_dict, orc_cvr64 *out);
/* the same with column phases (phases > 1; rows must have ascending columns: -6 otherwise) and a cap on the rows of a chunk */
int  orc_cvr64_build_ex(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, const void *vals, int is_f32,
                        int S, int64_t split_threshold, int use_dict, int phases, int64_t max_rows, orc_cvr64 *out);
/* the same with a hub table of at most hub_max entries (0 = none; not together with phases) */
int  orc_cvr64_build_hub(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, const void *vals, int is_f32,
                         int S, int64_t split_threshold, int use_dict, int phases, int64_t max_rows, int64_t hub_max, orc_cvr64 *out);
/* the same with narrow chunks (16-bit column offsets; -8 when a chunk spans too many columns or with dictionary / phases / hubs) */
int  orc_cvr64_build_all(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, const void *vals, int is_f32,
                         int S, int64_t split_threshold, int use_dict, int phases, int64_t max_rows, int64_t hub_max, int narrow, orc_cvr64 *out);
/* the same with the whole of x re-ordered by popularity (reorder != 0, needs hub_max > 0) */
int  orc_cvr64_build_full(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, const void *vals, int is_f32,
                          int S, int64_t split_threshold, int use_dict, int phases, int64_t max_rows, int64_t hub_max, int reorder, int narrow, orc_cvr64 *out);
/* the same with wide row tags (tag16 != 0, phases > 1 only): rows of the pieces in 16-bit tags of their own; and with the
 * (row, phase) segments cut into pieces of at most piece_max elements (piece_max > 0) */
int  orc_cvr64_build_tag(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, const void *vals, int is_f32,
                         int S, int64_t split_threshold, int use_dict, int phases, int64_t max_rows, int64_t hub_max, int reorder, int narrow, int tag16, int64_t piece_max, orc_cvr64 *out);
/* interleaved chunks (cvr_options.interleave): the chunk's non-zeros dealt to the lanes in column order, every slot a piece of its own */
int  orc_cvr64_build_ilv(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, const void *vals, int is_f32,
                         int S, int64_t split_threshold, int use_dict, int64_t max_rows, int tag16, orc_cvr64 *out);
/* gang chunks (cvr_options.gang; cvr_amd/csrc/cvr_format.h): the plan of interleaved chunks, but the `gang` consecutive chunks of a workgroup are sorted
 * TOGETHER -- element e of the gang's list sorted by (column, position) stands in group e / 256, step (e / 64) % 4, lane e % 64 of the gang's stream (its chunks'
 * allocations, one behind the other) --, a slot's tag = chunk inside the gang * ystage + row inside the chunk; without 16-bit tags the column word holds the
 * column's offset from its group's first column in bits [0, 17) and the tag above (-9: an offset does not fit 17 bits; the product then takes 16-bit tags) */
int  orc_cvr64_build_gang(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *cols, const void *vals, int is_f32,
                          int S, int64_t split_threshold, int use_dict, int64_t max_rows, int tag16, int gang, int ystage, orc_cvr64 *out);
void orc_cvr64_free(orc_cvr64 *c);
/* interpret the image exactly as the HIP kernel does (same per-lane order of operations) */
void orc_cvr64_spmv(const orc_cvr64 *c, const void *x, void *y);

#ifdef __cplusplus
}
#endif
#endif

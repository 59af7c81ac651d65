// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, container-only (needs /root/reference + AVX-512F).
//
// Links against the UNMODIFIED reference object (spmv.cpp compiled with -Dmain=cvr_ref_main,
// see oracle/Makefile target _ref/spmv_ref.o) and dumps what the reference computes, so that
// tests/golden/*.npz can pin oracle/cvr_oracle.c.  Nothing here is shipped or timed.
//
// It reproduces main()'s allocation contract (spmv.cpp:1777-1829):
//   y buffers numRows+1 (+slack, Q11), x numCols (+slack, Q1: 1-based columns read x[numCols]),
//   split zero-initialised, Nblock[t] = 1, record buffer 2*(numRows+240+32T) ints,
// runs the reference's CSR loop (spmv.cpp:1843-1850) on the reference loader's arrays itself,
// then pre_processing (spmv.cpp:565) and spmv_compute_kernel (spmv.cpp:1016) for the requested
// thread count, and writes little-endian raw arrays + a manifest into <outdir>.
//
// usage: ref_harness <file.mtx> <outdir> <T> [xmode: ones|rand]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <immintrin.h>

void readMatrix(char *filename, double **val_ptr, int **cols_ptr, int **rowDelimiters_ptr,
                int *n, int *numRows, int *numCols);
void pre_processing(int Nthrds, int N_start, int N_step, int *vPack_Nblock, int *vPack_vec_record,
                    int *vPack_nnz_rows, double *vPack_vec_vals, int *vPack_vec_cols, double *h_val,
                    int *h_cols, int *vPack_vec_final, int *vPack_vec_final_2, int *vPack_split,
                    double *refOut, int nItems, int numRows, int omega, int *h_rowDelimiters,
                    char *filename);
void spmv_compute_kernel(int Nthrds, int N_start, int N_step, int *vPack_Nblock,
                         int *vPack_vec_record, int *vPack_nnz_rows, double *vPack_vec_vals,
                         int *vPack_vec_cols, double *h_val, int *h_cols, int *vPack_vec_final,
                         int *vPack_vec_final_2, int *vPack_split, double *refOut, int nItems,
                         int numRows, int omega, int *h_rowDelimiters, char *filename,
                         double *h_vec, int Ntimes);

static void dump(const std::string &dir, const char *name, const void *p, size_t bytes) {
    std::string path = dir + "/" + name;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (bytes) fwrite(p, 1, bytes, f);
    fclose(f);
}

// splitmix64 -> uniform [-1, 1): the repo-wide seeded x, seed 0xC0FFEE, index j
static double x_rand(uint64_t j) {
    uint64_t z = 0xC0FFEEull + (j + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s file.mtx outdir T [ones|rand]\n", argv[0]); return 2; }
    char *fn = argv[1];
    std::string out = argv[2];
    int T = atoi(argv[3]);
    bool xr = argc > 4 && !strcmp(argv[4], "rand");
    const int SLACK = 64;

    double *val; int *cols, *rp; int n, nr, nc;
    readMatrix(fn, &val, &cols, &rp, &n, &nr, &nc);

    std::vector<double> x(nc + SLACK), yref(nr + SLACK, 0.0);
    for (int j = 0; j < nc + SLACK; j++) x[j] = xr ? x_rand(j) : 1.0;
    // the reference's CSR loop, spmv.cpp:1843-1850 (rows 0..numRows-1, j ascending)
    for (int i = 0; i < nr; i++) {
        double sum = 0;
        for (int j = rp[i]; j < rp[i + 1]; j++) sum += val[j] * x[cols[j]];
        yref[i] = sum;
    }

    double *y = (double *)_mm_malloc(sizeof(double) * (nr + SLACK), 64);
    for (int i = 0; i < nr + SLACK; i++) y[i] = 0;
    double *cv = (double *)_mm_malloc(sizeof(double) * n, 64);
    int *cc = (int *)_mm_malloc(sizeof(int) * n, 64);
    int *fin = (int *)_mm_malloc(sizeof(int) * 16 * T, 64);
    int *fin2 = (int *)_mm_malloc(sizeof(int) * 16 * T, 64);
    size_t nrec = 2 * ((size_t)nr + 240 + (size_t)T * 32);
    int *rec = (int *)_mm_malloc(sizeof(int) * nrec, 64);
    const int SENT = -0x7f7f7f7f;
    for (size_t i = 0; i < nrec; i++) rec[i] = SENT;
    for (int i = 0; i < 16 * T; i++) { fin[i] = SENT; fin2[i] = SENT; }
    int *split = (int *)calloc(2 * T, sizeof(int));
    int *nblock = (int *)malloc(sizeof(int) * T);
    for (int i = 0; i < T; i++) nblock[i] = 1;
    int *nnz_rows = (int *)_mm_malloc(sizeof(int) * 4 * T, 64);

    pre_processing(T, 0, T, nblock, rec, nnz_rows, cv, cc, val, cols, fin, fin2, split, y, n, nr, 1,
                   rp, fn);
    spmv_compute_kernel(T, 0, T, nblock, rec, nnz_rows, cv, cc, val, cols, fin, fin2, split, y, n,
                        nr, 1, rp, fn, x.data(), 1);

    dump(out, "csr_val.f64", val, sizeof(double) * n);
    dump(out, "csr_col.i32", cols, sizeof(int) * n);
    dump(out, "csr_rowptr.i32", rp, sizeof(int) * (nr + 2));
    dump(out, "x.f64", x.data(), sizeof(double) * (nc + 1));
    dump(out, "y_csr.f64", yref.data(), sizeof(double) * nr);
    dump(out, "y_cvr.f64", y, sizeof(double) * nr);
    dump(out, "cvr_val.f64", cv, sizeof(double) * n);
    dump(out, "cvr_col.i32", cc, sizeof(int) * n);
    dump(out, "cvr_record.i32", rec, sizeof(int) * nrec);
    dump(out, "cvr_split.i32", split, sizeof(int) * 2 * T);
    dump(out, "cvr_final2.i32", fin2, sizeof(int) * 16 * T);
    dump(out, "cvr_nnz_rows.i32", nnz_rows, sizeof(int) * 4 * T);
    std::string m = out + "/manifest.txt";
    FILE *f = fopen(m.c_str(), "w");
    fprintf(f, "nItems %d\nnumRows %d\nnumCols %d\nT %d\nxmode %s\nrecord_sentinel %d\n", n, nr, nc,
            T, xr ? "rand" : "ones", SENT);
    fclose(f);
    return 0;
}

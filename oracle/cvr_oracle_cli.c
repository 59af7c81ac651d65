/* oracle/cvr_oracle_cli.c -- TEST INFRASTRUCTURE: the CPU restatement behind the reference's CLI.
 *
 *   ./spmv_cvr_cpu [matrix.mtx] [nThreads] [nIters]          (argv contract: spmv.cpp:1693, 1703, 1771)
 *
 * This is BASELINE.json configs[0] ("reference CPU/OpenMP path on host cores, plumbing, no GPU") and
 * the `cpu_baseline` leg of bench.py.  It prints the reference's four greppable lines
 * (spmv.cpp:1009, 1662, 1664, 1932/1935; README.md:47-49) and one JSON line.  Timing follows the
 * reference (zeroing of y outside the timer, spmv.cpp:1026-1033) and also reports the figure with
 * the zeroing inside.
 */
#define _GNU_SOURCE
#include "cvr_oracle.h"
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s matrix.mtx nThreads nIters\n", argv[0]); return 2; }
    const char *fn = argv[1];
    int T = atoi(argv[2]), iters = atoi(argv[3]);
    if (T < 1) T = 1;
    if (iters < 1) iters = 1;
    orc_csr m;
    int rc = orc_read_matrix(fn, &m);
    if (rc) { fprintf(stderr, "Error: unable to read matrix file %s (%d)\n", fn, rc); return 1; }
    const int use_rand = getenv("CVR_X") && !strcmp(getenv("CVR_X"), "rand");
    double *x = (double *)malloc(sizeof(double) * ((size_t)m.numCols + 2));
    for (int j = 0; j < m.numCols + 2; j++) x[j] = use_rand ? orc_x_rand((uint64_t)j) : 1.0;  /* fill, spmv.cpp:556-563 */
    double *yref = (double *)calloc((size_t)m.numRows + 2, sizeof(double));
    double *y = (double *)calloc((size_t)m.numRows + 2, sizeof(double));
    orc_csr_spmv(m.numRows, m.rowptr, m.cols, m.val, x, yref);

    orc_cvr8 c;
    double t0 = now();
    rc = orc_cvr8_convert(&m, T, &c);
    double tpre = now() - t0;
    if (rc) { fprintf(stderr, "Error: matrix too small for %d chunks\n", T); return 1; }
    printf("The Pre-processing(CSR->CVR)   Time of CVR   is %g seconds.   [file: %s] [threads: %d]\n", tpre, fn, T);

    orc_cvr8_spmv(&c, x, y, T); /* warm-up */
    t0 = now();
    for (int k = 0; k < iters; k++) orc_cvr8_spmv(&c, x, y, T);
    double t = (now() - t0) / iters;
    printf("The SpMV Execution Time of CVR    is %g seconds.   [file: %s] [threads: %d]\n", t, fn, T);
    printf("         The Throughput of CVR    is %g GFlops.    [file: %s] [threads: %d]\n", m.nItems / t / 1e9, fn, T);

    long wrong = 0;                                                            /* spmv.cpp:1916-1938 */
    for (int i = 0; i < m.numRows; i++) { double d = fabs(y[i] - yref[i]); if (d * d > 0.000001) wrong++; }
    if (!wrong) printf("     Very Good! Your result is correct  \n");
    else printf("Warning: %ld out of %d is wrong\n", wrong, m.nItems);
    printf("{\"backend\":\"cpu-port\",\"threads\":%d,\"iters\":%d,\"nnz_padded\":%d,\"rows\":%d,\"preprocess_s\":%.6g,"
           "\"spmv_s\":%.6g,\"gflops_2nnz\":%.6g,\"wrong\":%ld}\n",
           T, iters, m.nItems, m.numRows, tpre, t, 2.0 * m.nItemsRaw / t / 1e9, wrong);
    return 0;
}

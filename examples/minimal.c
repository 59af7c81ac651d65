/* examples/minimal.c -- the whole boundary in one screen: a 4 x 4 CSR matrix, y = A x on the GPU through libcvr_amd.so.
 *   gcc -std=c99 -Iinclude examples/minimal.c -Lcvr_amd -lcvr_amd -Wl,-rpath,$PWD/cvr_amd -o minimal && ./minimal
 * (what the reference's main does around pre_processing / spmv_compute_kernel, spmv.cpp:1857, 1882) */
#include <stdio.h>
#include "cvr_amd.h"

int main(void)
{
    /* [ 1 . 2 . ]
     * [ . . . . ]      an empty row: written as 0 on every SpMV
     * [ 3 4 . 5 ]
     * [ . . 6 . ]                                                            */
    const int64_t row_ptr[5] = {0, 2, 2, 5, 6};
    const int32_t col_idx[6] = {0, 2, 0, 1, 3, 2};
    const double  vals[6] = {1, 2, 3, 4, 5, 6};
    const double  x[4] = {1, 10, 100, 1000};
    double        y[4] = {-1, -1, -1, -1};

    cvr_csr_view view = {4, 4, row_ptr, col_idx, vals, /*is_f32*/ 0, /*arrays_on_device*/ 0};
    cvr_options  opt;
    cvr_default_options(&opt);                 /* device 0, everything else chosen from the matrix */
    cvr_handle *h = NULL;
    double      pre_s = 0;
    cvr_timing  tm;
    if (cvr_create(&h, &view, &opt) || cvr_preprocess(h, 0, &pre_s) || cvr_spmv(h, x, y, /*iters*/ 3, &tm)) {
        fprintf(stderr, "cvr: %s\n", cvr_last_error());
        cvr_destroy(h);
        return 1;
    }
    printf("y = %g %g %g %g   (CSR->CVR %.3g s, %.3g s per SpMV)\n", y[0], y[1], y[2], y[3], pre_s, tm.mean_s);
    cvr_destroy(h);
    return !(y[0] == 201 && y[1] == 0 && y[2] == 5043 && y[3] == 600);
}

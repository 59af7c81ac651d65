// tools/ubench/seg_sort.hip -- what the gang converter's sort costs: rocprim::radix_sort_pairs over (gang << cbits | column) keys (the product's form) against
// rocprim::segmented_radix_sort_pairs over the column alone, one segment per gang (the elements of a gang are contiguous before the sort).
//   hipcc --offload-arch=gfx950 -O3 seg_sort.hip -o seg_sort ;  seg_sort [n=69000000] [segments=741] [cbits=19]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
int main(int argc, char **argv)
{
    const size_t   n = argc > 1 ? atol(argv[1]) : 69000000;
    const unsigned nseg = argc > 2 ? atoi(argv[2]) : 741, cbits = argc > 3 ? atoi(argv[3]) : 19;
    unsigned gbits = 1; while ((1u << gbits) < nseg) gbits++;
    std::vector<unsigned> key(n), keyc(n), val(n), off(nseg + 1);
    std::mt19937 rng(1);
    for (unsigned s = 0; s <= nseg; s++) off[s] = (unsigned)(n * (size_t)s / nseg);
    for (unsigned s = 0; s < nseg; s++) for (size_t i = off[s]; i < off[s + 1]; i++) { const unsigned c = rng() & ((1u << cbits) - 1); keyc[i] = c; key[i] = (s << cbits) | c; val[i] = (unsigned)i; }
    unsigned *dk, *dk2, *dv, *dv2, *doff; void *tmp = nullptr; size_t tb = 0, tb2 = 0;
    CK(hipMalloc(&dk, n * 4)); CK(hipMalloc(&dk2, n * 4)); CK(hipMalloc(&dv, n * 4)); CK(hipMalloc(&dv2, n * 4)); CK(hipMalloc(&doff, (nseg + 1) * 4));
    CK(hipMemcpy(doff, off.data(), (nseg + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, val.data(), n * 4, hipMemcpyHostToDevice));
    CK(rocprim::radix_sort_pairs(nullptr, tb, dk, dk2, dv, dv2, n, 0, cbits + gbits));
    CK(rocprim::segmented_radix_sort_pairs(nullptr, tb2, dk, dk2, dv, dv2, (unsigned)n, nseg, doff, doff + 1, 0, cbits));
    CK(hipMalloc(&tmp, tb > tb2 ? tb : tb2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned> ra(n), rb(n);
    for (int which = 0; which < 2; which++) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            CK(hipMemcpy(dk, which ? keyc.data() : key.data(), n * 4, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0));
            if (which == 0) CK(rocprim::radix_sort_pairs(tmp, tb, dk, dk2, dv, dv2, n, 0, cbits + gbits));
            else CK(rocprim::segmented_radix_sort_pairs(tmp, tb2, dk, dk2, dv, dv2, (unsigned)n, nseg, doff, doff + 1, 0, cbits));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipMemcpy((which ? rb : ra).data(), dv2, n * 4, hipMemcpyDeviceToHost));
        printf("%s: %.3f ms (n %zu, %u segments, %u key bits)\n", which ? "segmented_radix_sort_pairs (column only)" : "radix_sort_pairs (gang | column)    ", best, n, nseg, which ? cbits : cbits + gbits);
    }
    size_t diff = 0; for (size_t i = 0; i < n; i++) diff += ra[i] != rb[i];
    printf("permutations differ at %zu places (0 = the segmented sort is the same stable sort)\n", diff);
    return 0;
}

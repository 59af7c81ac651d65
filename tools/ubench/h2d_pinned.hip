// h2d_pinned.hip -- how fast does a large host array reach the device: pageable hipMemcpy against hipHostRegister +
// hipMemcpy + hipHostUnregister (the caller's CSR arrays are ordinary malloc memory)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 830, n = mb << 20;
    char *h = (char *)malloc(n);
    memset(h, 1, n);
    void *d; CK(hipMalloc(&d, n));
    CK(hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipHostRegister(h, n, hipHostRegisterDefault)); double t2 = now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double t3 = now();
        CK(hipHostUnregister(h)); double t4 = now();
        printf("%zu MB: pageable copy %.1f ms (%.1f GB/s) | register %.1f ms + pinned copy %.1f ms (%.1f GB/s) + unregister %.1f ms = %.1f ms\n", mb,
               (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n / (t3 - t2) / 1e9, (t4 - t3) * 1e3, (t4 - t1) * 1e3);
    }
    return 0;
}

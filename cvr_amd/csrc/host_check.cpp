// host_check.cpp -- sanitizer driver for the host-side sources (make asan-check): both loader modes, the binary image
// round trip, the keyed cache (miss / hit / stale), the planner at several chunk lengths and thresholds, the CSR loop, on every file given.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cvr_amd.h"
#include "cvr_plan.h"

// copy src to dst, with `tail` (may be NULL) appended; 0 on success
static int copy_file(const char *src, const char *dst, const char *tail)
{
    FILE *i = fopen(src, "rb"), *o = i ? fopen(dst, "wb") : nullptr;
    if (!i || !o) { if (i) fclose(i); return 1; }
    char   buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, i)) > 0) fwrite(buf, 1, n, o);
    if (tail) fputs(tail, o);
    fclose(i);
    fclose(o);
    return 0;
}

int main(int argc, char **argv)
{
    int failures = 0;
    for (int a = 1; a < argc; a++) {
        for (int mode = 0; mode < 2; mode++) {
            cvr_mm_matrix m;
            int           rc = cvr_mm_read(argv[a], mode, &m);
            if (rc) { printf("%s mode %d: rc %d\n", argv[a], mode, rc); if (mode == 0) failures++; continue; }
            std::vector<double> x((size_t)m.ncols + 1), y((size_t)m.nrows + 1);
            cvr_fill_x(x.data(), m.ncols, 1);
            cvr_csr_spmv_host(m.nrows, m.row_ptr, m.col_idx, m.vals, x.data(), y.data(), 2);
            for (int S : {4, 8, 32})
                for (long long thr : {0LL, 1LL, 1000000LL}) {
                    cvr::Plan p = cvr::plan_chunks(m.nrows, m.row_ptr, S, thr);
                    long long slots = 0;
                    for (const cvr::Chunk &c : p.chunks) slots += 64LL * S - c.pad_cnt;
                    long long want = 0;
                    for (long long r = 0; r < m.nrows; r++) { long long l = m.row_ptr[r + 1] - m.row_ptr[r]; want += l > 0 ? l : 1; }
                    if (slots != want) { printf("%s: planner lost slots (%lld vs %lld)\n", argv[a], slots, want); failures++; }
                    if ((long long)p.chunks.size() > cvr::plan_bound(m.nrows, m.nnz, S)) { printf("%s: plan_bound too small\n", argv[a]); failures++; }
                }
            const std::string bin = std::string("/tmp/host_check_") + std::to_string(a) + "_" + std::to_string(mode) + ".bin";
            cvr_mm_matrix     b;
            if (cvr_mm_write_bin(bin.c_str(), &m) || cvr_mm_read_bin(bin.c_str(), &b)) { printf("%s: binary image failed\n", argv[a]); failures++; }
            else {
                if (b.nnz != m.nnz || memcmp(b.row_ptr, m.row_ptr, sizeof(int64_t) * ((size_t)m.nrows + 1))) { printf("%s: binary image differs\n", argv[a]); failures++; }
                cvr_mm_free(&b);
            }
            remove(bin.c_str());
            // keyed cache beside a copy of the file: miss, hit, stale after the source grows by a comment line
            const std::string cp = std::string("/tmp/host_check_") + std::to_string(a) + "_" + std::to_string(mode) + ".mtx";
            if (copy_file(argv[a], cp.c_str(), nullptr) == 0) {
                cvr_mm_matrix c1, c2, c3;
                int           hit = -1;
                if (cvr_mm_read_cached(cp.c_str(), mode, &c1, &hit) || hit != 0) { printf("%s: first cached read (hit %d)\n", argv[a], hit); failures++; }
                else {
                    if (cvr_mm_read_cached(cp.c_str(), mode, &c2, &hit) || hit != 1) { printf("%s: second cached read (hit %d)\n", argv[a], hit); failures++; }
                    else {
                        if (c2.nnz != m.nnz || memcmp(c2.col_idx, m.col_idx, sizeof(*m.col_idx) * (size_t)m.nnz)) { printf("%s: cached image differs\n", argv[a]); failures++; }
                        cvr_mm_free(&c2);
                    }
                    copy_file(argv[a], cp.c_str(), "%% a comment line more\n");
                    if (cvr_mm_read_cached(cp.c_str(), mode, &c3, &hit) || hit != 0) { printf("%s: stale cache served (hit %d)\n", argv[a], hit); failures++; }
                    else cvr_mm_free(&c3);
                    cvr_mm_free(&c1);
                }
                remove(cp.c_str());
                remove((cp + ".ref.csrbin").c_str());
                remove((cp + ".strict.csrbin").c_str());
            }
            cvr_mm_free(&m);
        }
    }
    cvr_mm_matrix m;
    if (cvr_mm_read("/nonexistent", 0, &m) != CVR_ERR_IO) failures++;
    printf("host_check: %d file(s), %d failure(s)\n", argc - 1, failures);
    return failures ? 1 : 0;
}

import numpy as np, cvr_amd
for n in (64, 1000, 10000, 100000):
    rp = np.arange(n + 1, dtype=np.int64) * 4
    ci = (np.arange(4 * n) % n).astype(np.int32)
    va = np.ones(4 * n)
    A = cvr_amd.CvrMatrix(n, n, rp, ci, va)
    print(n, "rows:", A.info.nchunks, "chunks S", A.info.steps_per_chunk, "%.2f us per SpMV" % (A.bench(100, 2000) * 1e6))
    A.close()

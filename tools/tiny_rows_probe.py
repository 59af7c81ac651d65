import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import cvr_amd
n = 2_400_000
rng = np.random.default_rng(3)
rp = np.arange(n + 1, dtype=np.int64)
near = rng.random(n) < 0.5
ci = np.where(near, np.clip(np.arange(n) + rng.integers(-200, 200, n), 0, n - 1), rng.integers(0, n, n)).astype(np.int32)
va = rng.standard_normal(n)
t = time.time()
A = cvr_amd.CvrMatrix(n, n, rp, ci, va)
i = A.info
print("create %.2f s: fused %d S %d wpb %d win %d phases %d chunks %d image %.1f MB near %.2f" % (time.time() - t, i.preprocess_fused, i.steps_per_chunk, i.waves_per_block, i.x_window, i.col_phases, i.nchunks, i.image_bytes / 1e6, i.near_diagonal_share))
x = np.cos(np.arange(n) * 0.37)
y, _ = A.spmv(x)
ref = va * x[ci]
print("max err", np.abs(y - ref).max(), "us", A.bench(5, 20) * 1e6)

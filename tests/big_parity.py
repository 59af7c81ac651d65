#!/usr/bin/env python3
"""tests/big_parity.py [band<n>|livejournal|rmat<scale>] -- full-size parity + timing of one large matrix on one GPU:
y against the CSR oracle (all rows), run-to-run bit equality, preprocessing breakdown."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cvr_amd
import oraclelib as O
from cvr_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "livejournal"
t0 = time.time()
if name.startswith("band"):
    n, nc, rp, ci, va = synth.banded_sym(int(float(name[4:])))
elif name.startswith("rmat"):
    n, nc, rp, ci, va = synth.rmat(int(name[4:]), dtype=np.float32)
else:
    n, nc, rp, ci, va = synth.livejournal_like()
print(f"{name}: {n} x {nc}, nnz {len(ci)}, generated in {time.time() - t0:.1f}s", flush=True)
t0 = time.time()
A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
i = A.info
print(f"create+preprocess {time.time() - t0:.2f}s: S {i.steps_per_chunk}, chunks {i.nchunks}, rows cut {i.nshared}, column panels {i.col_panels}, dictionary {i.value_dict}, image {i.image_bytes / 1e9:.2f} GB, "
      f"plan {i.plan_s * 1e3:.1f} ms, upload {i.upload_s * 1e3:.1f} ms, convert {i.convert_s * 1e3:.2f} ms", flush=True)
x = synth.x_rand(nc, va.dtype)
y, _ = A.spmv(x)
y2, _ = A.spmv(x)
yref, absy = O.csr_spmv64(rp, ci, va, x)
tol = 1e-5 if va.dtype == np.float32 else 1e-12
bad, worst = O.tol_check(y, yref, absy + 1e-30, tol=tol)
s = A.bench(5, 20)
balg = synth.b_alg(n, nc, len(ci), va.dtype.itemsize)
print(f"rows off: {len(bad)} (worst rel {worst:.2e}); run-to-run bitwise equal: {bool(np.array_equal(y.view(np.uint8), y2.view(np.uint8)))}; "
      f"{s * 1e6:.1f} us/SpMV, {2 * len(ci) / s / 1e9:.1f} GFLOP/s, {balg / s / 1e9:.0f} GB/s algorithmic = {balg / s / 8e12 * 100:.1f}% of 8 TB/s")

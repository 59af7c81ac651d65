"""round 5 diagnostic: does synth_dev.rmat_rows(scale, lo, hi) on the GPU equal the slice of the matrix built in one piece, and is the build reproducible?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cvr_amd import synth_dev as D

scale = 20
n = 1 << scale
deg = D.rmat_row_degrees(scale, device="cuda")
bounds, grp = D.partition_from_degrees(deg, 8, 1250)
print("bounds", bounds.tolist())
whole = [t.cpu().numpy() for t in D.rmat_rows(scale, 0, n, device="cuda")]
whole2 = [t.cpu().numpy() for t in D.rmat_rows(scale, 0, n, device="cuda")]
print("whole reproducible:", [bool(np.array_equal(a, b)) for a, b in zip(whole, whole2)])
print("degrees == diff(rp):", bool(np.array_equal(np.diff(whole[0]), deg.cpu().numpy())))
for p in range(8):
    lo, hi = int(bounds[p]), int(bounds[p + 1])
    rp, ci, va = [t.cpu().numpy() for t in D.rmat_rows(scale, lo, hi, device="cuda")]
    a, b = int(whole[0][lo]), int(whole[0][hi])
    ok = [bool(np.array_equal(rp, whole[0][lo:hi + 1] - a)), bool(np.array_equal(ci, whole[1][a:b])), bool(np.array_equal(va, whole[2][a:b]))]
    print(f"shard {p}: rows [{lo}, {hi}) nnz {len(ci)} vs {b - a}: rp/ci/va equal {ok}")
    if not all(ok) and len(ci) == b - a:
        d = np.nonzero(ci != whole[1][a:b])[0]
        print("   first differing positions", d[:5], "of", len(d), "; rows of them", np.searchsorted(rp, d[:5], side="right") - 1)

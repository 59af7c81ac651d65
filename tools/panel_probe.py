#!/usr/bin/env python3
"""tools/panel_probe.py -- probe for a column-panelled layout: split the matrix into P column panels with equal nnz,
compact each panel's non-empty rows, time each panel's SpMV alone (its slice of x then fits the L2s) and sum."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "webgoogle"
plist = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
slist = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 32]
if name == "livejournal":
    n, nc, rp, ci, va = synth.livejournal_like()
elif name.startswith("rmat"):
    n, nc, rp, ci, va = synth.rmat(int(name[4:]), dtype=np.float64)
else:
    n, nc, rp, ci, va = synth.web_google_like()
nnz = len(ci)
rows = np.repeat(np.arange(n), np.diff(rp))
balg = synth.b_alg(n, nc, nnz)
for P in plist:
    cc = np.bincount(ci, minlength=nc).cumsum()
    bounds = np.concatenate([[0], np.searchsorted(cc, np.arange(1, P) * nnz / P), [nc]])
    panel = np.searchsorted(bounds[1:-1], ci, side="right")
    for S in slist:
        tot, sub = 0.0, 0
        for j in range(P):
            m = panel == j
            r, c, v = rows[m], ci[m], va[m]
            ur, inv = np.unique(r, return_inverse=True)
            lrp = np.zeros(len(ur) + 1, dtype=np.int64)
            lrp[1:] = np.cumsum(np.bincount(inv, minlength=len(ur)))
            A = cvr_amd.CvrMatrix(len(ur), nc, lrp, c, v, steps_per_chunk=S, x_window=0)
            A.spmv(synth.x_rand(nc))
            tot += A.bench(5, 30)
            sub += len(ur)
            A.close()
        print(f"P={P} S={S}: sum of panel SpMVs {tot * 1e6:7.2f} us, sub-rows {sub} ({sub / nnz:.3f} nnz), z bytes {sub * 8 / 1e6:.1f} MB", flush=True)

#!/usr/bin/env python3
"""three computing + three helper wavefronts per workgroup against the automatic four + two on the hold-out shapes whose images get helpers (stream > 192 MB)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import holdout as H
import cvr_amd
from cvr_amd import synth
for name, P in (("lj_half", 8), ("orkut_half", 8), ("uniform16", 8)):
    n, nc, rp, ci, va = H.SHAPES[name]()
    x = synth.x_rand(nc, va.dtype)
    yref = cvr_amd.csr_spmv_host(rp, ci, va.astype(np.float64), x.astype(np.float64), nthreads=16)
    absy = cvr_amd.csr_spmv_host(rp, ci, np.abs(va).astype(np.float64), np.abs(x.astype(np.float64)), nthreads=16)
    out = []
    for rep in range(1):
        for label, dbg, kw in (("4 + 2 (automatic)", "", {}), ("3 + 3", "ilv_helpers=3", dict(waves_per_block=3, col_panels=P, interleave=1)),
                               ("3 + 2", "ilv_helpers=2", dict(waves_per_block=3, col_panels=P, interleave=1)), ("4 + 3", "ilv_helpers=3", dict(waves_per_block=4, col_panels=P, interleave=1)),
                               ("3 + 4", "ilv_helpers=4", dict(waves_per_block=3, col_panels=P, interleave=1)), ("2 + 3", "ilv_helpers=3", dict(waves_per_block=2, col_panels=P, interleave=1))):
            os.environ["CVR_DEBUG"] = dbg
            A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
            y, _ = A.spmv(x)
            wrong = int(np.count_nonzero(np.abs(y.astype(np.float64) - yref) > 1e-12 * absy + 1e-300))
            s = A.bench(10, 100)
            i = A.info
            out.append(f"{label}: {s * 1e6:.1f} us (S {i.steps_per_chunk} wpb {i.waves_per_block} P {i.col_panels} wrong {wrong})")
            A.close()
    print(f"{name:12s} " + " | ".join(out), flush=True)

#!/usr/bin/env python3
"""from which size on do helper wavefronts pay?  soc-LiveJournal1 shape at several scales, automatic layout, helpers forced off / 2 per chunk; image and partial-sum sizes beside"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvr_amd
from cvr_amd import synth
for scale in (0.4, 0.45, 0.5):
    n, nc, rp, ci, va = synth.livejournal_like(scale=scale, seed=5)
    out = []
    for h in (0, 2, -1):
        os.environ["CVR_DEBUG"] = f"ilv_helpers={h},ilv_flip={1 if h else 0}" if h >= 0 else "fused_trace"
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
        s = A.bench(10, 100)
        i = A.info
        out.append(f"helpers {h if h >= 0 else 'automatic'}: {s * 1e6:.1f} us")
        A.close()
    print(f"scale {scale}: rows {n} nnz {len(ci)} image {i.image_bytes / 1e6:.0f} MB panels {i.col_panels} ilv {i.interleave} S {i.steps_per_chunk} x {nc * 8 / 1e6:.0f} MB | " + " | ".join(out), flush=True)

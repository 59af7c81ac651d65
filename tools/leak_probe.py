#!/usr/bin/env python3
"""tools/leak_probe.py -- device memory before / after many create + destroy cycles on the paths that give an attempt up (one-submission
preprocessing not confirmed / more chunks than workgroup slots) and on the batched panel plans: the free memory must come back."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth

def tiny_rows():
    n = 2_400_000
    rng = np.random.default_rng(3)
    rp = np.arange(n + 1, dtype=np.int64)
    near = rng.random(n) < 0.5
    ci = np.where(near, np.clip(np.arange(n) + rng.integers(-200, 200, n), 0, n - 1), rng.integers(0, n, n)).astype(np.int32)
    return n, n, rp, ci, rng.standard_normal(n)

cases = {"resident (one submission)": synth.web_google_like(0.5)[:5], "band (probe does not confirm)": synth.banded_sym(400000, 13)[:5],
         "tiny rows (too many chunks)": tiny_rows(), "panels (batched plans)": synth.livejournal_like(0.125)[:5]}
torch.cuda.init()
for name, (n, nc, rp, ci, va) in cases.items():
    kw = dict(col_panels=4, hub_table=0) if name.startswith("panels") else {}
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw); A.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
        A.spmv(np.ones(nc))
        A.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print("%-32s free before %8.1f MB, after 20 cycles %8.1f MB, difference %6.1f MB" % (name, free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6), flush=True)

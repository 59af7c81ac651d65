#!/usr/bin/env python3
"""tools/rmat_reorder_probe.py [scale] [f32|f64] -- R-MAT with the hub table: x re-ordered by popularity or not, the matrix stream cached or in
uncached memory (CVR_STREAM_UNCACHED=1 in the environment of the call)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth, synth_dev as D

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
f32 = (sys.argv[2] if len(sys.argv) > 2 else "f32") == "f32"
dev = torch.device("cuda", 0)
n = 1 << scale
lrp, lci, lva = D.rmat_rows(scale, 0, n, device=dev)
if not f32:
    lva = lva.double()
torch.cuda.synchronize()
for reorder in (0, 1):
    A = cvr_amd.CvrMatrix.from_device(n, n, lrp.data_ptr(), lci.data_ptr(), lva.data_ptr(), is_f32=f32, hub_reorder=reorder, keep_csr=True)
    i = A.info
    t = A.bench(5, 30)
    print(f"R-MAT-{scale} {'fp32' if f32 else 'fp64'} uncached stream {os.environ.get('CVR_STREAM_UNCACHED', '0')} hub_reorder {reorder}: hub entries {i.hub_entries} reorder {i.hub_reorder} panels {i.col_panels}: {t * 1e6:8.1f} us", flush=True)
    A.close()

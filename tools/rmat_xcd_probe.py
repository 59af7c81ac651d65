#!/usr/bin/env python3
"""tools/rmat_xcd_probe.py [scale] -- R-MAT (fp32, built on the GPU): the library's own choice against column panels dealt to the XCDs
without hub tables (8 / 16 / 32 panels): how much the column skew of R-MAT unbalances equal-width panels."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth, synth_dev as D

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dev = torch.device("cuda", 0)
n = 1 << scale
lrp, lci, lva = D.rmat_rows(scale, 0, n, device=dev)
torch.cuda.synchronize()
nnz = int(lrp[-1])
print(f"R-MAT-{scale}: {n} rows, {nnz} non-zeros, fp32", flush=True)
x = synth.x_rand(n, np.float32)
for panels, hub, S in [(-1, -1, 0), (8, 0, 0), (16, 0, 0), (32, 0, 0), (16, 0, 16), (32, 0, 16), (64, 0, 16)]:
    try:
        A = cvr_amd.CvrMatrix.from_device(n, n, lrp.data_ptr(), lci.data_ptr(), lva.data_ptr(), is_f32=True, col_panels=panels, hub_table=hub, steps_per_chunk=S, keep_csr=True)
    except Exception as e:      # noqa: BLE001
        print(f"panels {panels} hub {hub} S {S}: {e}", flush=True)
        continue
    i = A.info
    t = A.bench(5, 30)
    print(f"panels asked {panels:3d} hub_table {hub:2d}: panels {i.col_panels:3d} launches {i.spmv_launches} S {i.steps_per_chunk} hub entries {i.hub_entries} chunks {i.nchunks}: {t * 1e6:8.1f} us", flush=True)
    A.close()

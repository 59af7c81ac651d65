#!/usr/bin/env python3
"""tools/rmat_create_timing.py [scale] -- cvr_create + cvr_preprocess of an R-MAT matrix (fp32, built on the GPU) with CVR_CREATE_TIMING=1:
where the preprocessing of a panelled matrix with hub tables spends its time"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth_dev as D
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
n = 1 << scale
lrp, lci, lva = D.rmat_rows(scale, 0, n, device=dev)
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    A = cvr_amd.CvrMatrix.from_device(n, n, lrp.data_ptr(), lci.data_ptr(), lva.data_ptr(), is_f32=True)
    i = A.info
    print("create + preprocess %.1f ms: plan %.1f hub %.1f dict %.1f preprocess wall %.1f; panels %d hub entries %d" % ((time.perf_counter() - t) * 1e3, i.plan_s * 1e3, i.hub_select_s * 1e3, i.dict_s * 1e3, i.preprocess_wall_s * 1e3, i.col_panels, i.hub_entries), flush=True)
    A.close()

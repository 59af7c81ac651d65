"""kernel time of one rank's shard of the web-Google-shaped matrix for N = 1, 2, 4, 8 row shards: the default layout of
cvr_create, the plain layout (one chunk per workgroup, S by rule) and the layout cvr_tune measures
(PYTHONPATH=. python tools/shard_kernel_probe.py) -- what the strong-scaling curve of bench.py has to work with"""
import os
import numpy as np
import cvr_amd
from cvr_amd import shard, synth

nrows, ncols, rp, ci, va = synth.web_google_like(1.0)[:5]
for n in (1, 2, 4, 8):
    bounds = shard.row_partition(rp, n)
    lrows, lrp, lci, lva = shard.local_csr(rp, ci, va, bounds, 0)
    line = []
    for label, kw in (("auto", {}), ("plain", dict(waves_per_block=1, x_window=0, col_phases=1)), ("tuned", dict(tune_steps=True))):
        A = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, **kw)
        i = A.info
        t = A.bench(50, 1000) * 1e6
        line.append(f"{label}: {t:.2f} us (S {i.steps_per_chunk}, {i.waves_per_block} chunks/wg, window {i.x_window}, phases {i.col_phases}, {i.nchunks} chunks"
                    + (f", tuning {A.tuning_s * 1e3:.0f} ms" if label == "tuned" else "") + ")")
        A.close()
    print(f"N={n} rank 0: rows {lrows} nnz {int(lrp[-1])} | " + " | ".join(line), flush=True)

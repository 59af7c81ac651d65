"""tools/r03_rule_probe.py -- the automatic layout against cvr_tune on web-Google-shaped matrices of several sizes and on row shards
(PYTHONPATH=. python tools/r03_rule_probe.py; CVR_RESIDENT_RULE=old for the former rule)"""
import numpy as np
import cvr_amd
from cvr_amd import shard, synth
for scale in (1.0, 0.7, 0.5, 0.35, 0.25, 0.125):
    n, nc, rp, ci, va = synth.web_google_like(scale)[:5]
    line = []
    for label, kw in (("auto", {}), ("tuned", dict(tune_steps=True))):
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw); i = A.info
        line.append(f"{label} {A.bench(50, 800) * 1e6:6.2f} us (S {i.steps_per_chunk} w {i.waves_per_block} win {i.x_window} P {i.col_phases})")
        A.close()
    print(f"scale {scale:5.3f} rows {n:7d} nnz {len(ci):8d}: " + " | ".join(line), flush=True)
nrows, ncols, rp, ci, va = synth.web_google_like(1.0)[:5]
for N in (2, 4, 8):
    bounds = shard.row_partition(rp, N)
    lrows, lrp, lci, lva = shard.local_csr(rp, ci, va, bounds, 0)
    line = []
    for label, kw in (("auto", {}), ("tuned", dict(tune_steps=True))):
        A = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, **kw); i = A.info
        line.append(f"{label} {A.bench(50, 800) * 1e6:6.2f} us (S {i.steps_per_chunk} w {i.waves_per_block} win {i.x_window} P {i.col_phases})")
        A.close()
    print(f"shard 1/{N} rows {lrows:7d} nnz {int(lrp[-1]):8d}: " + " | ".join(line), flush=True)

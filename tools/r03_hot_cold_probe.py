#!/usr/bin/env python3
"""tools/r03_hot_cold_probe.py [scale] [f32|f64] -- R-MAT split by column POPULARITY: the columns ranked by their counts, A_hot = the entries in the H most
popular columns (their slice of x fits every XCD's L2) as one image over the whole chip, A_cold = the rest as column panels, one per XCD at a time.
A prototype with the existing kernels (the split is made here, with torch): t_hot + t_cold against the single image with hub table (+ re-ordered x).
PYTHONPATH=. python tools/r03_hot_cold_probe.py 22 f64"""
import sys
import torch
import cvr_amd
from cvr_amd import synth_dev as D

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
f32 = (sys.argv[2] if len(sys.argv) > 2 else "f64") == "f32"
dev = torch.device("cuda", 0)
n = 1 << scale
rp, ci, va = D.rmat_rows(scale, 0, n, device=dev)
if not f32:
    va = va.double()
A = cvr_amd.CvrMatrix.from_device(n, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), is_f32=f32); i = A.info
print(f"R-MAT-{scale} {'fp32' if f32 else 'fp64'} as the rules build it: {A.bench(5, 50) * 1e6:8.1f} us (hub entries {i.hub_entries} reorder {i.hub_reorder} panels {i.col_panels})", flush=True)
A.close()
cnt = torch.bincount(ci.long(), minlength=n)
order = torch.argsort(cnt, descending=True)
rank = torch.empty_like(order); rank[order] = torch.arange(n, device=dev)
cr = rank[ci.long()].int()                                  # column = popularity rank
rows = torch.repeat_interleave(torch.arange(n, device=dev), (rp[1:] - rp[:-1]))
for H in (131072, 262144, 524288, 1048576):
    out, tot = [], 0.0
    for name, m in (("hot", cr < H), ("cold", cr >= H)):
        c2, v2, r2 = cr[m].contiguous(), va[m].contiguous(), rows[m]
        rp2 = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp2[1:] = torch.cumsum(torch.bincount(r2, minlength=n), 0)
        torch.cuda.synchronize()
        for kw in ((dict(col_panels=1, hub_table=0),) if name == "hot" else (dict(col_panels=8, hub_table=0), dict(col_panels=16, hub_table=0))):
            B = cvr_amd.CvrMatrix.from_device(n, n, rp2.data_ptr(), c2.data_ptr(), v2.data_ptr(), is_f32=f32, **kw); j = B.info
            t = B.bench(5, 50) * 1e6
            out.append(f"{name} {int(m.sum()) / len(ci) * 100:.0f} % panels {j.col_panels}: {t:7.1f}")
            B.close()
            if name == "hot": tot += t
            else: best = min(t, best) if kw["col_panels"] == 16 else t
    print(f"H {H}: " + " | ".join(out) + f" | hot + best cold: {tot + best:7.1f} us", flush=True)

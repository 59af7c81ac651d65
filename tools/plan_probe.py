"""tools/plan_probe.py -- seconds of the host planner and of the device planner on the bench shapes' row pointers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cvr_amd
from cvr_amd import synth

for name, S, mr in (("web_google_like", 48, 1170), ("web_google_like", 32, 0), ("livejournal_like", 32, 0)):
    nrows, ncols, rp, ci, va = getattr(synth, name)()[:5]
    for rep in range(2):
        r = cvr_amd.plan_selfcheck(rp, S, 0, mr)
    print(f"{name:12s} rows {nrows:9d} S {S} max_rows {mr}: host {r['host_s']*1e6:9.1f} us   device {r['device_s']*1e6:9.1f} us   chunks {r['nchunks']}")

# one column panel of the LiveJournal shape (its sub-rows: the rows' entries inside the panel's column range)
nrows, ncols, rp, ci, va = synth.livejournal_like()[:5]
W = (ncols + 8) // 9
rows = np.repeat(np.arange(nrows), np.diff(rp))
for p in (0, 4):
    inp = (ci // W) == p
    cnt = np.bincount(rows[inp], minlength=nrows)
    lens = cnt[cnt > 0]
    prp = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=prp[1:])
    for rep in range(2):
        r = cvr_amd.plan_selfcheck(prp, 32, 0, 0)
    print(f"LJ panel {p}: sub-rows {len(lens)} nnz {prp[-1]} longest {lens.max()} rows>512 {(lens > 512).sum()}: host {r['host_s']*1e6:9.1f} us   device {r['device_s']*1e6:9.1f} us   chunks {r['nchunks']}")

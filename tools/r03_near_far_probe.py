"""tools/r03_near_far_probe.py -- soc-LiveJournal1 shape split into A_near (|col - row| <= D: a band, one plain image) and A_far (the rest: column panels,
one per XCD at a time), each through the library as it is: the sum of their SpMV times against the whole matrix (16 panels).  A prototype of the
near / far form with the EXISTING kernels (every gather still goes to an L2); the split is made on the host here.
(PYTHONPATH=. python tools/r03_near_far_probe.py)"""
import numpy as np
import scipy.sparse as sp
import cvr_amd
from cvr_amd import synth

n, nc, rp, ci, va = synth.livejournal_like()[:5]
rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
A = cvr_amd.CvrMatrix(n, nc, rp, ci, va); i = A.info
print(f"whole: {A.bench(10, 100) * 1e6:7.1f} us (panels {i.col_panels})", flush=True)
A.close()
for D in (25_000, 50_000, 100_000, 200_000):
    near = np.abs(ci.astype(np.int64) - rows) <= D
    out = []
    tot = 0.0
    for name, m in (("near", near), ("far", ~near)):
        M = sp.csr_matrix((va[m], ci[m], np.concatenate([[0], np.cumsum(np.bincount(rows[m], minlength=n))])), shape=(n, nc))
        for P in ((1,) if name == "near" else (-1, 16)):
            B = cvr_amd.CvrMatrix(n, nc, M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data, col_panels=P); j = B.info
            t = B.bench(10, 100) * 1e6
            out.append(f"{name} ({m.sum() / len(ci) * 100:.0f} % of nnz) panels {j.col_panels} S {j.steps_per_chunk} w {j.waves_per_block}: {t:6.1f} us")
            B.close()
            if name == "near" or P == 16:
                tot += t
    print(f"D {D}: " + " | ".join(out) + f" | near + far(16): {tot:6.1f} us", flush=True)

"""tools/r03_mid_size_probe.py -- matrices between the resident layout's range and the panel rule's 24 MB of x: automatic choice against one image
and against eight column panels, one per XCD; y of the automatic choice checked against a CSR product on the host
(PYTHONPATH=. python tools/r03_mid_size_probe.py [quick])"""
import sys
import numpy as np
import scipy.sparse as sp
import cvr_amd
from cvr_amd import synth


def run(name, n, nc, rp, ci, va):
    line = []
    for P in (-1, 1, 8):
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=P); i = A.info
        t = A.bench(20, 300) * 1e6
        line.append(f"{'auto' if P < 0 else f'panels {P}'}: {t:7.2f} us (S {i.steps_per_chunk} w {i.waves_per_block} P {i.col_phases} panels {i.col_panels} hub {i.hub_entries})")
        if P < 0:
            x = synth.x_rand(nc, va.dtype)
            y = A.spmv(x)[0]
            ref = sp.csr_matrix((va.astype(np.float64), ci, rp), shape=(n, nc)) @ x.astype(np.float64)
            err = float(np.max(np.abs(y - ref)) / max(1e-30, float(np.max(np.abs(ref)))))
            assert err < (2e-5 if va.dtype == np.float32 else 1e-12), err
        A.close()
    print(f"{name}: rows {n} nnz {len(ci)} x {nc * va.dtype.itemsize / 1e6:.1f} MB {va.dtype} | " + " | ".join(line), flush=True)


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for scale in ((2.0, 2.4, 3.2) if quick else (1.6, 2.0, 2.2, 2.4, 2.8, 3.2, 3.6)):
    run(f"web-Google shape x {scale}", *synth.web_google_like(scale)[:5])
run("LiveJournal shape x 0.5", *synth.livejournal_like(0.5)[:5])
n, nc, rp, ci, va = synth.rmat(22, dtype=np.float32)[:5]
run("R-MAT-22 fp32", n, nc, rp, ci, va)
n, nc, rp, ci, va = synth.rmat(21, dtype=np.float64)[:5]
run("R-MAT-21 fp64", n, nc, rp, ci, va)
n, nc, rp, ci, va = synth.web_google_like(2.8)[:5]
run("web-Google shape x 2.8 fp32 (x 10 MB: below the range)", n, nc, rp, ci, va.astype(np.float32))

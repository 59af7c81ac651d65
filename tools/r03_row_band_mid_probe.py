"""tools/r03_row_band_mid_probe.py -- web-Google shapes beyond one resident pass (x of 16-26 MB) as ROW BANDS: the rows cut into B bands of equal
slots, each built with the automatic rules as a matrix of its own over all of x (a resident launch with column phases and window where the
rules choose them); the sum of the bands' SpMV times against the whole matrix (eight column panels).
(PYTHONPATH=. python tools/r03_row_band_mid_probe.py)"""
import numpy as np
import cvr_amd
from cvr_amd import synth

for scale in (2.4, 2.8, 3.2, 3.6):
    n, nc, rp, ci, va = synth.web_google_like(scale)[:5]
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va); i = A.info
    whole = A.bench(20, 300) * 1e6
    wl = f"whole: {whole:7.2f} us (panels {i.col_panels})"
    A.close()
    slots = np.concatenate([[0], np.cumsum(np.maximum(np.diff(rp), 1))])
    for B in (2, 3):
        cuts = np.searchsorted(slots, np.linspace(0, slots[-1], B + 1)); cuts[0], cuts[-1] = 0, n
        tot, desc = 0.0, []
        for b in range(B):
            r0, r1 = int(cuts[b]), int(cuts[b + 1])
            Ab = cvr_amd.CvrMatrix(r1 - r0, nc, rp[r0:r1 + 1], ci, va, col_panels=1); j = Ab.info
            t = Ab.bench(20, 300) * 1e6
            tot += t
            desc.append(f"{t:.1f} (S {j.steps_per_chunk} w {j.waves_per_block} P {j.col_phases} shared {j.nshared})")
            Ab.close()
        wl += f" | {B} bands: {tot:7.2f} us = " + " + ".join(desc)
    print(f"scale {scale}: nnz {len(ci)} x {nc * 8 / 1e6:.1f} MB | {wl}", flush=True)

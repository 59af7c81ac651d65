"""tools/r03_long_resident_probe.py -- web-Google-shaped matrices of 2-4 times the headline's size (x of 15-30 MB): the automatic layout
against the resident layout with chunks longer than the rule's 128 steps (PYTHONPATH=. python tools/r03_long_resident_probe.py)"""
import math
import numpy as np
import cvr_amd
from cvr_amd import synth
for scale in (2.0, 3.0, 4.0):
    n, nc, rp, ci, va = synth.web_google_like(scale)[:5]
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va); i = A.info
    print(f"scale {scale}: rows {n} nnz {len(ci)} x {nc * 8 / 1e6:.1f} MB | auto {A.bench(20, 300) * 1e6:7.2f} us (S {i.steps_per_chunk} w {i.waves_per_block} win {i.x_window} P {i.col_phases} panels {i.col_panels})", flush=True)
    A.close()
    slots = (len(ci) + n / 4) * 1.006
    for w in (8, 6):
        S = math.ceil(slots / (64 * w * 252) / 4) * 4
        for P in (min(64, max(2, round(nc * 8 / 450e3))), 32, 16):
            try:
                A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, steps_per_chunk=S, waves_per_block=w, x_window=12288, col_phases=P); i = A.info
                print(f"    resident S {S} w {w} P {P}: {A.bench(20, 300) * 1e6:7.2f} us (chunks {i.nchunks}, piece_max {i.piece_max}, tags16 {i.row_tags16})", flush=True)
                A.close()
            except Exception as e:      # noqa: BLE001
                print(f"    resident S {S} w {w} P {P}: {e}", flush=True)

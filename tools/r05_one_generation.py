#!/usr/bin/env python3
"""more wavefronts per workgroup so that a panel's chunks fit ONE generation of workgroups per XCD: soc-LiveJournal1 shape x 0.45 / 0.5 / 0.6 / 0.7, 8 panels interleaved"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvr_amd
from cvr_amd import synth
for scale in (0.45, 0.5, 0.6, 0.7):
    n, nc, rp, ci, va = synth.livejournal_like(scale=scale, seed=5)
    out = []
    for wpb, S in ((0, 0), (5, 0), (5, 576), (6, 0), (6, 576), (8, 0), (8, 576)):
        os.environ["CVR_DEBUG"] = ""
        try:
            A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **(dict(col_panels=8, interleave=1, waves_per_block=wpb, steps_per_chunk=S) if wpb else {}))
        except Exception as e:
            out.append(f"wpb {wpb} S {S}: {str(e)[:30]}"); continue
        s = A.bench(10, 100)
        i = A.info
        out.append(f"wpb {i.waves_per_block} S {i.steps_per_chunk} P {i.col_panels} chunks {i.nchunks} ({i.nchunks / i.col_panels * (i.col_panels / 8) / i.waves_per_block / 32:.2f} generations): {s * 1e6:.1f} us")
        A.close()
    print(f"scale {scale}: nnz {len(ci)} | " + " | ".join(out), flush=True)

#!/usr/bin/env python3
"""is the step between the soc-LiveJournal1 shape x 0.4 (89 us) and x 0.45 (139 us) the chunk length (544 / 256 steps: one / two generations of workgroups per XCD) or the Infinity Cache?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cvr_amd
from cvr_amd import synth
for scale in (0.4, 0.45, 0.5):
    n, nc, rp, ci, va = synth.livejournal_like(scale=scale, seed=5)
    out = []
    for S in (0, 256, 384, 544, 576):
        for h in (0, 2):
            os.environ["CVR_DEBUG"] = f"ilv_helpers={h},ilv_flip={1 if h else 0}"
            try:
                A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=8, interleave=1, steps_per_chunk=S)
            except Exception as e:
                out.append(f"S {S}: {str(e)[:40]}"); continue
            s = A.bench(10, 100)
            i = A.info
            out.append(f"S {i.steps_per_chunk} ({i.nchunks} chunks) helpers {h}: {s * 1e6:.1f}")
            A.close()
    print(f"scale {scale}: nnz {len(ci)} | " + " | ".join(out), flush=True)

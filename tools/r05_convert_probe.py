#!/usr/bin/env python3
"""tools/r05_convert_probe.py -- the interleaved converter's stages (CVR_DEBUG=ilv_clocks) on the soc-LiveJournal1 shape for several chunk
lengths (pairs per thread of the in-LDS sort) and radix widths; conversion time from the device events beside them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvr_amd
from cvr_amd import synth
n, nc, rp, ci, va = synth.livejournal_like()
x = synth.x_rand(nc)
cvr_amd.CvrMatrix(n, nc, rp, ci, va).close()
for S, dbg in ((0, ""), (0, "ilv_rb10"), (0, "ilv_rb6"), (256, ""), (128, ""), (508, ""), (576, "")):
    os.environ["CVR_DEBUG"] = "ilv_clocks" + ("," + dbg if dbg else "")
    sys.stderr.write(f"--- steps_per_chunk {S or 'automatic'} {dbg}\n"); sys.stderr.flush()
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, steps_per_chunk=S, col_panels=16, interleave=1)
    i = A.info
    t = A.bench(5, 30)
    sys.stderr.write(f"    S {i.steps_per_chunk} chunks {i.nchunks}: convert {i.convert_s * 1e3:.2f} ms (device events), plan {i.plan_s * 1e3:.2f} ms, SpMV {t * 1e6:.1f} us\n"); sys.stderr.flush()
    A.close()

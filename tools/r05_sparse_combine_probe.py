#!/usr/bin/env python3
"""the combine pass of mostly-empty matrices with larger workgroups (CVR_DEBUG=combine_batch=9..12) on the wiki-Talk shape, its x 2 and the forum-like hold-out shape"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import holdout as H
import cvr_amd
from cvr_amd import synth, synth_dev as D
def wiki():
    n, rp, ci, va = D.wikitalk_like(device="cuda")
    return n, n, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()
for name, gen in (("wikitalk", wiki), ("wikitalk_x2", H.SHAPES["wikitalk_x2"]), ("forum_sparse", H.SHAPES["forum_sparse"])):
    n, nc, rp, ci, va = gen()
    out = []
    for cb in ("", "9", "10", "11", "12", "", "9"):
        os.environ["CVR_DEBUG"] = f"combine_batch={cb}" if cb else ""
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
        s = A.bench(20, 200)
        out.append(f"{cb or 'default'}: {s * 1e6:.2f}")
        A.close()
    print(f"{name:13s} " + " | ".join(out), flush=True)

#!/usr/bin/env python3
"""tools/wpb_probe.py -- interleaved column panels with four (the rule) against two wavefronts per workgroup on shapes whose chunks come out short
(round 5: the wiki-Talk shape ran 40.0 us with 4 x 92 steps, 36.7 us with 2 x 176): where does the longer chunk pay?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: F401,E402
import cvr_amd  # noqa: E402
from cvr_amd import synth  # noqa: E402
from cvr_amd import synth_dev as D  # noqa: E402
import holdout as H  # noqa: E402


def shapes():
    n, rp, ci, va = D.wikitalk_like(device="cuda")
    yield "wikitalk", (n, n, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy())
    yield "wikitalk_x2", H.wikitalk_x2()
    for f in (2.2, 3.0):
        yield f"webgoogle x{f}", synth.power_law_graph(int(916_428 * f), int(5_105_039 * f), 0.193, 456, 20261002)
    yield "lj_half", H.lj_half()
    yield "uniform16", H.uniform16()
    yield "orkut_half", H.orkut_half()


def main():
    for name, (n, nc, rp, ci, va) in shapes():
        nnz = len(ci)
        balg = synth.b_alg(n, nc, nnz, va.dtype.itemsize)
        x = synth.x_rand(nc, va.dtype)
        yref = cvr_amd.csr_spmv_host(rp, ci, va.astype(np.float64), x.astype(np.float64), nthreads=16)
        absy = cvr_amd.csr_spmv_host(rp, ci, np.abs(va).astype(np.float64), np.abs(x.astype(np.float64)), nthreads=16)
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
        i = A.info
        P, ilv = i.col_panels, i.interleave
        res = []
        for label, kw in (("automatic", {}), ("2 wavefronts", dict(col_panels=P, interleave=1, waves_per_block=2)), ("3 wavefronts", dict(col_panels=P, interleave=1, waves_per_block=3))):
            if label != "automatic":
                if not ilv:
                    continue
                A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
            y, _ = A.spmv(x)
            wrong = int(np.count_nonzero(np.abs(y.astype(np.float64) - yref) > 1e-12 * absy + 1e-300))
            t = min(A.bench(10, 100) for _ in range(2))
            j = A.info
            res.append(f"{label}: {t * 1e6:8.2f} us ({balg / t / 8e12 * 100:4.1f} %) S {j.steps_per_chunk} wpb {j.waves_per_block} P {j.col_panels} chunks {j.nchunks} wrong {wrong}")
            A.close()
        print(f"{name:16s} nnz {nnz:10d} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/layout_probe.py MATRIX 'k=v,k=v' ... -- SpMV time of one stand-in under explicit layout options (CvrMatrix keyword arguments), each
checked against the library's host CSR loop; the first line is the library's own choice.
  MATRIX: webgoogle | livejournal | orkut | wikitalk | rmat<scale>"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth


def load(name):
    if name == "webgoogle":
        return synth.web_google_like()
    if name == "livejournal":
        return synth.livejournal_like()
    if name.startswith("rmat"):
        return synth.rmat(int(name[4:]), dtype=np.float32)
    from cvr_amd import synth_dev as D
    n, rp, ci, va = (D.orkut_like if name == "orkut" else D.wikitalk_like)(device="cuda")
    return n, n, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()


def main():
    name = sys.argv[1]
    n, nc, rp, ci, va = load(name)
    nnz = len(ci)
    balg = synth.b_alg(n, nc, nnz, va.dtype.itemsize)
    x = synth.x_rand(nc, va.dtype)
    xh = x.astype(np.float64)
    yref = cvr_amd.csr_spmv_host(rp, ci, va.astype(np.float64), xh, nthreads=16)
    absy = cvr_amd.csr_spmv_host(rp, ci, np.abs(va).astype(np.float64), np.abs(xh), nthreads=16)
    tol = 1e-5 if va.dtype == np.float32 else 1e-12
    print(f"# {name}: {n} x {nc}, nnz {nnz}, B_alg {balg / 1e6:.1f} MB", flush=True)
    for spec in [""] + sys.argv[2:]:
        kw = {}
        for item in filter(None, spec.split(",")):
            k, v = item.split("=")
            kw[k] = int(v)
        try:
            t0 = time.perf_counter()
            A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
            wall = time.perf_counter() - t0
        except Exception as e:
            print(f"  {spec or '(default)':50s} {e}", flush=True)
            continue
        y, _ = A.spmv(x)
        wrong = int(np.count_nonzero(np.abs(y.astype(np.float64) - yref) > tol * absy + 1e-300))
        s = A.bench(20, 200)
        i = A.info
        print(f"  {spec or '(default)':50s} {s * 1e6:9.2f} us  {balg / s / 8e12 * 100:5.1f} %  wrong {wrong}  S {i.steps_per_chunk} wpb {i.waves_per_block} panels {i.col_panels} ilv {i.interleave} "
              f"phases {i.col_phases} win {i.x_window} hub {i.hub_entries} ({i.hub_share:.2f}) reorder {i.hub_reorder} dict {i.value_dict} launches {i.spmv_launches} chunks {i.nchunks} rowcap {i.chunk_row_cap} image_MB {i.image_bytes / 1e6:.0f} create_ms {wall * 1e3:.1f}", flush=True)
        A.close()


if __name__ == "__main__":
    main()

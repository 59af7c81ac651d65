#!/usr/bin/env python3
"""tools/helper_probe.py MATRIX 'helpers,ahead,per_line[,k=v,...]' ... -- the interleaved kernel with HELPER wavefronts (scalar prefetch of the matrix
stream into the L2: cvr_spmv.hip, spmv_ilv_kernel) under explicit settings of the three knobs (CVR_DEBUG=ilv_helpers / ilv_ahead / ilv_per_line at
creation) and optional layout options, each run checked against the library's host CSR loop.  MATRIX as tools/layout_probe.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cvr_amd  # noqa: E402
from cvr_amd import synth  # noqa: E402
from layout_probe import load  # noqa: E402


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
    base = os.environ.get("CVR_DEBUG", "")
    print(f"# {name}: {n} x {nc}, nnz {nnz}, B_alg {balg / 1e6:.1f} MB", flush=True)
    for spec in sys.argv[2:]:
        parts = spec.split(",")
        h, a, p = (int(v) for v in parts[:3])
        kw = {k: int(v) for k, v in (item.split("=") for item in parts[3:]) if not k.startswith("dbg_")}
        dbg = [f"{k[4:]}={v}" for k, v in (item.split("=") for item in parts[3:]) if k.startswith("dbg_")]          # (dbg_name=value: another CVR_DEBUG knob)
        os.environ["CVR_DEBUG"] = ",".join(filter(None, [base, f"ilv_helpers={h}", f"ilv_ahead={a}", f"ilv_per_line={p}"] + dbg))
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
        y, _ = A.spmv(x)
        wrong = int(np.count_nonzero(np.abs(y.astype(np.float64) - yref) > tol * absy + 1e-300))
        s = A.bench(20, 200)
        i = A.info
        print(f"  helpers {h} ahead {a} per_line {p} {str(kw) + ' ' + ' '.join(dbg):40s} {s * 1e6:9.2f} us  {balg / s / 8e12 * 100:5.1f} %  wrong {wrong}  S {i.steps_per_chunk} wpb {i.waves_per_block} panels {i.col_panels} ilv {i.interleave} chunks {i.nchunks}", flush=True)
        A.close()


if __name__ == "__main__":
    main()

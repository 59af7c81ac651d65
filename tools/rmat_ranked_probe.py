#!/usr/bin/env python3
"""tools/rmat_ranked_probe.py SCALE 'k=v,..' ... -- R-MAT with its columns RENUMBERED by popularity on the host (most popular first, every row
re-sorted by the new index), handed to the library as an ordinary matrix: what column phases + an LDS window over the re-ordered x would run
at, before any of it is built into the preprocessing (the x[perm] pass in front of every SpMV, ~30 us at scale 22, is not in these times)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth


def main():
    scale = int(sys.argv[1])
    n, nc, rp, ci, va = synth.rmat(scale, dtype=np.float32)
    nnz = len(ci)
    cnt = np.bincount(ci, minlength=nc)
    order = np.argsort(-cnt, kind="stable")
    rank = np.empty(nc, dtype=np.int64)
    rank[order] = np.arange(nc)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    cr = rank[ci]
    o = np.lexsort((cr, rows))
    ci2, va2 = cr[o].astype(np.int32), va[o]
    top = np.cumsum(cnt[order])
    print(f"# rmat{scale}: nnz {nnz}; share of the non-zeros in the 36 000 / 262 144 / 1 048 576 most popular columns: {top[35999] / nnz:.3f} / {top[262143] / nnz:.3f} / {top[min(1048575, nc - 1)] / nnz:.3f}", flush=True)
    balg = synth.b_alg(n, nc, nnz, 4)
    for label, cols, vals in (("natural", ci, va), ("ranked", ci2, va2)):
        x = synth.x_rand(nc, np.float32)
        yref = cvr_amd.csr_spmv_host(rp, cols, vals.astype(np.float64), x.astype(np.float64), nthreads=16)
        absy = cvr_amd.csr_spmv_host(rp, cols, np.abs(vals).astype(np.float64), np.abs(x).astype(np.float64), nthreads=16)
        for spec in ([""] if label == "natural" else [""] + sys.argv[2:]):
            kw = {k: int(v) for k, v in (item.split("=") for item in filter(None, spec.split(",")))}
            try:
                A = cvr_amd.CvrMatrix(n, nc, rp, cols, vals, **kw)
            except Exception as e:
                print(f"  {label:8s} {spec or '(default)':60s} {e}", flush=True)
                continue
            y, _ = A.spmv(x)
            wrong = int(np.count_nonzero(np.abs(y.astype(np.float64) - yref) > 1e-5 * absy + 1e-300))
            s = A.bench(10, 100)
            i = A.info
            print(f"  {label:8s} {spec or '(default)':60s} {s * 1e6:9.2f} us {balg / s / 8e12 * 100:5.1f} %  wrong {wrong}  S {i.steps_per_chunk} wpb {i.waves_per_block} panels {i.col_panels} ilv {i.interleave} "
                  f"phases {i.col_phases} win {i.x_window} hub {i.hub_entries} ({i.hub_share:.2f}) reorder {i.hub_reorder} lds {i.lds_bytes} chunks {i.nchunks}", flush=True)
            A.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/sweep.py -- time the SpMV kernel over the format / launch knobs on one GPU (HIP events on the handle's
stream, cvr_spmv_bench).  Usage: python tools/sweep.py [webgoogle|livejournal|rmat22] [--S 8,16,32] ..."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("matrix", nargs="?", default="webgoogle")
    ap.add_argument("--S", default="8,16,32,64,128", help="0 = auto")
    ap.add_argument("--swz", default="1,0")
    ap.add_argument("--nt", default="0", help="stream run-ahead: 0|1 = one group beyond the gather, 2 = three")
    ap.add_argument("--thr", default="0")
    ap.add_argument("--depth", default="1")
    ap.add_argument("--win", default="-1", help="x window values staged in LDS per workgroup (-1 auto, 0 off)")
    ap.add_argument("--panels", default="-1", help="column panels (-1 auto, 1 off)")
    ap.add_argument("--wpb", default="0", help="wavefronts (consecutive chunks) per SpMV workgroup (0 = default 1)")
    ap.add_argument("--dict", default="-1", help="value dictionary (-1 auto, 0 off)")
    ap.add_argument("--phases", default="0", help="column phases (0 / 1 off, -1 auto)")
    ap.add_argument("--hub", default="0", help="hub table entries (0 off, -1 auto)")
    ap.add_argument("--narrow", default="-1", help="16-bit column offsets for narrow chunks (-1 auto, 0 off)")
    ap.add_argument("--check", action="store_true", help="compare y of every configuration with the host CSR loop")
    ap.add_argument("--colmask", default="0", help="comma list of hex masks: folds the x gather onto a small table (timing only)")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--fold", type=int, default=0, help="timing experiment: fold the column indices onto 2^k columns before building (x then fits the L2s)")
    a = ap.parse_args()
    t0 = time.time()
    if a.matrix == "webgoogle":
        n, nc, rp, ci, va = synth.web_google_like(scale=a.scale)
    elif a.matrix == "livejournal":
        n, nc, rp, ci, va = synth.livejournal_like(scale=a.scale)
    elif a.matrix.startswith("rmat64_"):
        n, nc, rp, ci, va = synth.rmat(int(a.matrix[7:]), dtype=np.float64)
    elif a.matrix.startswith("rmat"):
        n, nc, rp, ci, va = synth.rmat(int(a.matrix[4:]), dtype=np.float32)
    elif a.matrix.startswith("band"):
        n, nc, rp, ci, va = synth.banded_sym(int(float(a.matrix[4:])))
    else:
        raise SystemExit("unknown matrix")
    if a.fold:
        ci = (ci & ((1 << a.fold) - 1)).astype(np.int32)
        rows = np.repeat(np.arange(n), np.diff(rp))
        o = np.lexsort((ci, rows))                     # keep the columns of every row ascending (column phases need it)
        ci, va = ci[o], va[o]
    nnz = len(ci)
    vb = va.dtype.itemsize
    balg = synth.b_alg(n, nc, nnz, vb)
    print(f"# {a.matrix}: {n} x {nc}, nnz {nnz}, B_alg {balg / 1e6:.1f} MB, generated in {time.time() - t0:.1f}s", flush=True)
    yref = absy = None
    print("#   S  swz nt    thr  chunks   cut  slots/nnz   conv_us    us/spmv   GFLOP/s   GB/s(alg)  %8TB/s")
    for S in [int(s) for s in a.S.split(",")]:
        for thr in [int(s) for s in a.thr.split(",")]:
            for swz in [int(s) for s in a.swz.split(",")]:
                for (nt, cm, wpb, dp, win, pan, vd, php, hub, nar) in [(int(s), int(m, 16), int(wp), int(dp), int(w), int(pn), int(vd), int(php), int(hb), int(na)) for s in a.nt.split(",")
                                                   for m in a.colmask.split(",") for wp in a.wpb.split(",")
                                                   for dp in a.depth.split(",") for w in a.win.split(",") for pn in a.panels.split(",")
                                                   for vd in a.dict.split(",") for php in a.phases.split(",") for hb in a.hub.split(",") for na in a.narrow.split(",")]:
                    try:
                        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, steps_per_chunk=S, split_threshold=thr, xcd_swizzle=swz,
                                              debug_col_mask=cm, x_window=win, col_panels=pan, waves_per_block=wpb, value_dict=vd, col_phases=php, hub_table=hub, narrow_cols=nar)
                    except Exception as e:
                        print(f"  {S:4d} wpb {wpb} win {win}: {e}", flush=True)
                        continue
                    x = synth.x_rand(nc, va.dtype)
                    y, _ = A.spmv(x)
                    wrong = ""
                    if a.check and not cm:
                        if yref is None:
                            xh = x.astype(np.float64)
                            yref = cvr_amd.csr_spmv_host(rp, ci, va.astype(np.float64), xh, nthreads=8)
                            absy = cvr_amd.csr_spmv_host(rp, ci, np.abs(va).astype(np.float64), np.abs(xh), nthreads=8)
                        tol = 1e-5 if va.dtype == np.float32 else 1e-12
                        wrong = f"  wrong {int(np.count_nonzero(np.abs(y.astype(np.float64) - yref) > tol * absy + 1e-300))}"
                    s = A.bench(a.warmup, a.iters)
                    i = A.info
                    print(f"  {S:4d}  {swz}  {nt:2d}  {thr:6d}  {i.nchunks:6d} {i.nshared:5d}  {i.nslots / max(nnz, 1):8.4f}  {i.convert_s * 1e6:9.1f}  "
                          f"{s * 1e6:9.2f}  {2 * nnz / s / 1e9:8.1f}  {balg / s / 1e9:9.1f}  {balg / s / 8e12 * 100:6.1f}" + f"  depth {dp} wpb {wpb} win {i.x_window} panels {i.col_panels} dict {i.value_dict} phases {i.col_phases} hub {i.hub_entries} ({i.hub_share:.2f}, {i.hub_select_s * 1e3:.1f} ms) lds {i.lds_bytes} narrow {i.narrow_cols} reorder {i.hub_reorder} image_MB {i.image_bytes / 1e6:.0f} pre_wall_us {i.preprocess_wall_s * 1e6:.0f} plan_us {i.plan_s * 1e6:.0f} probe_us {i.probe_s * 1e6:.0f}{wrong}" + (f"  colmask {cm:#x}" if cm else ""), flush=True)
                    A.close()


if __name__ == "__main__":
    main()

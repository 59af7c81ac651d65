#!/usr/bin/env python3
"""tools/band_probe.py -- one ROW BAND of a large matrix as a resident launch: rows [r0, r1) x all columns with long chunks
(one per wavefront, 256 workgroups at once), the rows' sums in LDS and column phases over the whole of x.
Usage: python tools/band_probe.py livejournal --bands 2 --wpb 8,16 --phases 12,16,24 [--check]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cvr_amd
from cvr_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("matrix", nargs="?", default="livejournal")
    ap.add_argument("--bands", type=int, default=2)
    ap.add_argument("--wpb", default="8")
    ap.add_argument("--phases", default="16")
    ap.add_argument("--depth", default="1")
    ap.add_argument("--pmax", default="8", help="longest piece of a lane stream (0 = whole segments)")
    ap.add_argument("--slack", type=float, default=1.04, help="slots per chunk beyond the band's share (row caps pad chunks)")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--local-frac", type=float, default=-1.0, help="livejournal: share of the non-zeros drawn near the diagonal (default: the shape's 0.5)")
    ap.add_argument("--iters", type=int, default=100)
    a = ap.parse_args()
    t0 = time.time()
    if a.matrix == "livejournal" and a.local_frac >= 0:
        p = synth.LIVEJOURNAL
        n, nc, rp, ci, va = synth.power_law_graph(p["n"], p["nnz"], p["empty_frac"], p["max_deg"], p["seed"], alpha=1.9, local_frac=a.local_frac, local_scale=50000.0)
    elif a.matrix == "livejournal":
        n, nc, rp, ci, va = synth.livejournal_like()
    elif a.matrix.startswith("rmat64_"):
        n, nc, rp, ci, va = synth.rmat(int(a.matrix[7:]), dtype=np.float64)
    elif a.matrix.startswith("rmat"):
        n, nc, rp, ci, va = synth.rmat(int(a.matrix[4:]), dtype=np.float32)
    else:
        n, nc, rp, ci, va = synth.web_google_like()
    nnz = len(ci)
    vb = va.dtype.itemsize
    print(f"# {a.matrix}: {n} x {nc}, nnz {nnz}, generated in {time.time() - t0:.1f}s", flush=True)
    slots = np.concatenate([[0], np.cumsum(np.maximum(np.diff(rp), 1))])
    cuts = np.searchsorted(slots, np.linspace(0, slots[-1], a.bands + 1))
    cuts[0], cuts[-1] = 0, n
    x = synth.x_rand(nc, va.dtype)
    if a.check:
        import oraclelib as O
        yref, absy = O.csr_spmv64(rp, ci, va, x)
    print("# band rows nnz  wpb S phases depth | chunks tags rowcap slots/nnz lds  us  Greq/s(nnz/t)  wrong")
    for wpb in [int(v) for v in a.wpb.split(",")]:
        for P in [int(v) for v in a.phases.split(",")]:
            for dp, pm in [(int(v), int(w)) for v in a.depth.split(",") for w in a.pmax.split(",")]:
                tot = 0.0
                for b in range(a.bands):
                    r0, r1 = int(cuts[b]), int(cuts[b + 1])
                    bslots = int(slots[r1] - slots[r0])
                    S = int(np.ceil(bslots * a.slack / (256 * wpb * 64) / 4.0)) * 4
                    lrp = rp[r0:r1 + 1]
                    try:
                        A = cvr_amd.CvrMatrix(r1 - r0, nc, lrp, ci, va, steps_per_chunk=S, waves_per_block=wpb, x_window=0, col_phases=P, col_panels=1,
                                              split_threshold=32 * S, hub_table=0, piece_max=pm)
                    except Exception as e:
                        print(f"  band {b} wpb {wpb} S {S} P {P}: {e}")
                        continue
                    i = A.info
                    y, _ = A.spmv(x)
                    s = A.bench(5, a.iters)
                    tot += s
                    wrong = ""
                    if a.check:
                        bad, worst = O.tol_check(y, yref[r0:r1], absy[r0:r1], tol=1e-5 if vb == 4 else 1e-12)
                        wrong = f"wrong {len(bad)} worst {worst:.1e}"
                    bn = int(lrp[-1] - lrp[0])
                    print(f"  {b} {r1 - r0} {bn}  {wpb} {S} {P} {dp} pmax {i.piece_max} | {i.nchunks} {i.row_tags16} {i.chunk_row_cap} {i.nslots / max(bn, 1):.3f} {i.lds_bytes}  {s * 1e6:8.1f}  {bn / s / 1e9:6.1f}  {wrong}  pre {i.preprocess_wall_s * 1e3:.0f} ms plan {i.plan_s * 1e3:.0f} ms", flush=True)
                    A.close()
                print(f"# wpb {wpb} phases {P} depth {dp} pmax {pm}: {tot * 1e6:.1f} us over {a.bands} bands = {synth.b_alg(n, nc, nnz, vb) / tot / 8e12 * 100:.1f} % of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()

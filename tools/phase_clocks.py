#!/usr/bin/env python3
"""tools/phase_clocks.py [MATRIX] -- where the headline kernel's time goes (round-4 verdict, item 3): the SpMV kernel of the resident layout
(spmv_seg_kernel) built with per-wavefront time stamps (CVR_DEBUG=phase_clocks: the chip's 100-MHz real-time counter at entry / prologue
done / window barrier passed / loop done / rows stored), one launch, dumped as a per-XCD table:

  start    = a wavefront's entry after the launch's first entry (dispatch skew)
  prologue = entry -> first gather (descriptor and stream loads issued, accumulators zeroed, dictionary copied)
  loop before the window barrier = what a wavefront does between its prologue and the barrier (nothing in the product: the barrier stands in
             front of the first group); wait = at that barrier, for the loader wavefronts' LDS-direct loads of the 96-KiB window
  loop     = the gather loop, the wait taken out
  store    = rows written out (until the stores have left the wavefront)
  end      = a wavefront's last stamp after the launch's first entry; the kernel's duration is the largest + the launch floor's tail

MATRIX: webgoogle (default).  Output -> stdout (tools/r05 scripts copy it to profiles/r05_phase_clocks_*.txt)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVR_DEBUG"] = (os.environ.get("CVR_DEBUG", "") + ",phase_clocks").strip(",")
import cvr_amd  # noqa: E402
from cvr_amd import synth  # noqa: E402


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else float("nan")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "webgoogle"
    n, nc, rp, ci, va = synth.web_google_like() if name == "webgoogle" else getattr(synth, name)()
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
    i = A.info
    x = synth.x_rand(nc, va.dtype)
    A.spmv(x)
    t = A.bench(20, 50)          # (the stamped build; the product's build is what bench.py times)
    pc = A.phase_clocks().astype(np.int64)
    kind = pc[:, :, 7]
    comp = pc[kind == 1]
    load = pc[kind == 2]
    t0 = min(comp[:, 0].min(), load[:, 0].min() if len(load) else comp[:, 0].min())
    tick = 10.0          # ns per tick of the 100-MHz counter
    us = lambda v: v * tick / 1e3  # noqa: E731
    print(f"# {name}: {n} rows, {len(ci)} nnz; layout S {i.steps_per_chunk} x {i.waves_per_block} chunks per workgroup, window {i.x_window}, phases {i.col_phases}, dict {i.value_dict}; "
          f"{i.nchunks} chunks; stamped kernel {t * 1e6:.2f} us per SpMV (event-timed, back to back)")
    print(f"# {len(comp)} computing wavefronts, {len(load)} loader wavefronts; 100-MHz counter: 10 ns per tick")
    rows = [("start (entry - first entry)", comp[:, 0] - t0), ("prologue", comp[:, 1] - comp[:, 0]), ("loop before the window barrier", comp[:, 6] - comp[:, 1]),
            ("wait at the window barrier", comp[:, 2] - comp[:, 6]), ("loop behind the barrier", comp[:, 3] - comp[:, 2]), ("loop in all (without the wait)", comp[:, 3] - comp[:, 1] - (comp[:, 2] - comp[:, 6])),
            ("store", comp[:, 4] - comp[:, 3]), ("end (last stamp - first entry)", comp[:, 4] - t0)]
    print(f"{'phase, us':34s} {'min':>7s} {'p10':>7s} {'median':>7s} {'p90':>7s} {'max':>7s} {'mean':>7s}")
    for label, v in rows:
        v = us(v)
        print(f"{label:34s} {v.min():7.2f} {pct(v, 10):7.2f} {pct(v, 50):7.2f} {pct(v, 90):7.2f} {v.max():7.2f} {v.mean():7.2f}")
    if len(load):
        v = us(load[:, 1] - load[:, 0])
        print(f"{'loader: entry -> loads landed':34s} {v.min():7.2f} {pct(v, 10):7.2f} {pct(v, 50):7.2f} {pct(v, 90):7.2f} {v.max():7.2f} {v.mean():7.2f}")
    print("# per XCD (XCC id): workgroups' computing wavefronts -- median / max of loop and of end")
    xcc = comp[:, 5] & 0xf
    for xid in sorted(set(xcc.tolist())):
        m = xcc == xid
        lp, en, st = us(comp[m, 3] - comp[m, 1] - (comp[m, 2] - comp[m, 6])), us(comp[m, 4] - t0), us(comp[m, 0] - t0)
        wt, bf = us(comp[m, 2] - comp[m, 6]), us(comp[m, 6] - comp[m, 1])
        print(f"  XCC {xid}: {int(m.sum()):5d} wavefronts  start median {pct(st, 50):6.2f} | before barrier median {pct(bf, 50):6.2f} | barrier wait median {pct(wt, 50):6.2f} max {wt.max():6.2f} | "
              f"loop median {pct(lp, 50):6.2f} max {lp.max():6.2f} | end median {pct(en, 50):6.2f} max {en.max():6.2f}")
    end = us(comp[:, 4] - t0)
    loop_share = float(us(comp[:, 3] - comp[:, 1] - (comp[:, 2] - comp[:, 6])).mean() / end.max())
    print(f"# slowest wavefront ends at {end.max():.2f} us, the median one at {pct(end, 50):.2f} us; mean loop / slowest end = {loop_share:.2f}")
    A.close()


if __name__ == "__main__":
    main()

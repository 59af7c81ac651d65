"""tools/gang_check.py -- parity check of gang chunks (cvr_options.gang) on the GPU -- y against the CSR oracle, bitwise reruns -- on the small cases of the
parity tests (forced: one image, four or two wavefronts per workgroup, with and without 16-bit tags, fp32, no dictionary) and on scaled stand-ins
with the automatic layout"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cvr_amd
from cvr_amd import synth
import oraclelib as O
from test_gpu_parity import CASES

bad = 0
def check(tag, nrows, ncols, rp, ci, va, **kw):
    global bad
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, **kw)
    i = A.info
    f32 = np.asarray(va).dtype == np.float32
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode).astype(np.asarray(va).dtype)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        y2, _ = A.spmv(x)
        err = np.abs(y.astype(np.float64) - yref)
        tol = (1e-5 if f32 else 1e-12) * absy + 1e-300
        nb = int(np.count_nonzero(err > tol))
        same = bool(np.array_equal(y, y2))
        if nb or not same:
            bad += 1
        print(f"{tag:40s} {mode:4s} gang {i.gang} ilv {i.interleave} panels {i.col_panels} wpb {i.waves_per_block} S {i.steps_per_chunk} tags {i.row_tags16} dict {i.value_dict} chunks {i.nchunks} cut rows {i.nshared}: wrong {nb} rerun {'same' if same else 'DIFFERENT'}", flush=True)
    A.close()

for name in sorted(CASES):
    nrows, ncols, rp, ci, va = CASES[name]
    for S, wpb, tags in ((4, 4, -1), (16, 4, 1), (32, 2, -1), (64, 4, -1), (16, 8, -1)):
        try:
            check(f"{name} S{S} w{wpb} t{tags}", nrows, ncols, rp, ci, va, steps_per_chunk=S, waves_per_block=wpb, row_tags16=tags, interleave=1, gang=1, col_panels=1)
        except Exception as e:
            bad += 1
            print(f"{name} S{S} w{wpb} t{tags}: EXCEPTION {e!r}", flush=True)
n, nc, rp, ci, va = synth.web_google_like(scale=0.06, seed=5)
check("webgoogle x0.06 S508", n, nc, rp, ci, va, steps_per_chunk=508, waves_per_block=4, interleave=1, gang=1, col_panels=1)
check("webgoogle x0.06 S64 nodict", n, nc, rp, ci, va, steps_per_chunk=64, waves_per_block=4, interleave=1, gang=1, col_panels=1, value_dict=0)
check("webgoogle x0.06 S64 f32", n, nc, rp, ci, va.astype(np.float32), steps_per_chunk=64, waves_per_block=4, interleave=1, gang=1, col_panels=1, value_dict=0)
n, nc, rp, ci, va = synth.livejournal_like(scale=0.03)
check("lj x0.03 16 panels", n, nc, rp, ci, va, col_panels=16, interleave=1, gang=1)
check("lj x0.03 16 panels gang off", n, nc, rp, ci, va, col_panels=16, interleave=1, gang=0)
n, nc, rp, ci, va = synth.livejournal_like(scale=0.2)
check("lj x0.2 auto", n, nc, rp, ci, va)
print("FAILED" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)

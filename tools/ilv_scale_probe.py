import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import cvr_amd
from cvr_amd import synth
import oraclelib as O
sc = float(sys.argv[1]); P = int(sys.argv[2]); W = int(sys.argv[3]) if len(sys.argv) > 3 else 0; S = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n, nc, rp, ci, va = synth.livejournal_like(scale=sc)
A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=P, interleave=1, waves_per_block=W, steps_per_chunk=S)
i = A.info
print("scale", sc, "P", i.col_panels, "S", i.steps_per_chunk, "chunks", i.nchunks, "launches", i.spmv_launches, "tags", i.row_tags16, "wpb", i.waves_per_block, "lds", i.lds_bytes, flush=True)
x = synth.x_rand(nc)
y, t = A.spmv(x, iters=30)
yref, absy = O.csr_spmv64(rp, ci, va, x)
bad, worst = O.tol_check(y, yref, absy, tol=1e-12)
print("bad", len(bad), "worst", worst, "us", t.mean_s * 1e6, flush=True)

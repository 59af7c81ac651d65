"""seconds per iteration of the native power iteration on the web-Google-shaped matrix, one GPU and the sharded form with a
1-rank RCCL communicator (PYTHONPATH=. python tools/power_probe.py)"""
import numpy as np
import torch
import cvr_amd
from cvr_amd import power, synth

nrows, ncols, rp, ci, va = synth.web_google_like(1.0)[:5]
va = np.abs(va) + 0.5
A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
power.power_iteration(A, nrows, iters=20)
lam, x, sec = power.power_iteration(A, nrows, iters=200)
print("one GPU: lambda %.12g, %.2f us per iteration (SpMV alone %.2f us)" % (lam, sec * 1e6, A.bench(20, 200) * 1e6))
comm = cvr_amd.Comm(cvr_amd.comm_unique_id(), 1, 0, 0)
power.power_iteration(A, nrows, bounds=[0, nrows], comm=comm, iters=20)
lam2, x2, sec2 = power.power_iteration(A, nrows, bounds=[0, nrows], comm=comm, iters=200)
print("sharded form, 1 rank (all-gather every iteration, the gathered y read in place): lambda %.12g, %.2f us per iteration; same bits: %s" % (lam2, sec2 * 1e6, bool(torch.equal(x.view(torch.int64), x2.view(torch.int64)))))

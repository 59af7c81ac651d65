"""kernel time of one rank's shard of the web-Google-shaped matrix for N = 1, 2, 4, 8 row shards, default S and a sweep
(PYTHONPATH=. python tools/shard_kernel_probe.py) -- what the strong-scaling curve of bench.py has to work with"""
import sys
import numpy as np
import cvr_amd
from cvr_amd import shard, synth

DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nrows, ncols, rp, ci, va = synth.web_google_like(1.0)[:5]
print('depth', DEPTH)
for n in (1, 2, 4, 8):
    bounds = shard.row_partition(rp, n)
    for r in (0,):
        lrows, lrp, lci, lva = shard.local_csr(rp, ci, va, bounds, r)
        line = []
        cand = {0} | set(range(24, 68, 4)) | {72, 80, 84, 96, 112, 128, 168}
        for S in sorted(cand):
            A = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, steps_per_chunk=S, depth=DEPTH)
            t = A.bench(50, 1000) * 1e6
            line.append("%d%s:%.1f(%d)" % (A.info.steps_per_chunk, "*" if S == 0 else "", t, A.info.nchunks))
            A.close()
        A = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, tune_steps=True, depth=DEPTH)
        line.append("tuned -> S=%d: %.1f us, tuning %.1f ms" % (A.info.steps_per_chunk, A.bench(50, 1000) * 1e6, A.tuning_s * 1e3))
        A.close()
        print("N=%d rank %d rows %d nnz %d | " % (n, r, lrows, int(lrp[-1])) + " | ".join(line), flush=True)

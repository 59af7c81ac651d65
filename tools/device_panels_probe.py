"""cvr_create + cvr_preprocess wall time of the soc-LiveJournal1-shaped matrix with column panels: host arrays (host split,
upload of the split arrays) against device-resident arrays (split on the device)  (PYTHONPATH=. python tools/device_panels_probe.py)"""
import time
import numpy as np
import torch
import cvr_amd
from cvr_amd import synth

n, nc, rp, ci, va = synth.livejournal_like(1.0)[:5]
dev = torch.device("cuda", 0)
trp, tci, tva = torch.from_numpy(rp).to(dev), torch.from_numpy(np.ascontiguousarray(ci, dtype=np.int32)).to(dev), torch.from_numpy(va).to(dev)
torch.cuda.synchronize()
x = synth.x_rand(nc)
for rep in range(2):
    t0 = time.perf_counter(); A = cvr_amd.CvrMatrix(n, nc, rp, ci, va); t1 = time.perf_counter()
    B = cvr_amd.CvrMatrix.from_device(n, nc, trp.data_ptr(), tci.data_ptr(), tva.data_ptr()); t2 = time.perf_counter()
    same = np.array_equal(A.spmv(x)[0].view(np.uint8), B.spmv(x)[0].view(np.uint8))
    print("host arrays: %.0f ms (%d panels, plan %.0f ms, upload %.0f ms) | device arrays: %.0f ms (%d panels, plan %.0f ms, upload %.0f ms) | same y bits: %s; %.1f / %.1f us per SpMV" % (
        (t1 - t0) * 1e3, A.info.col_panels, A.info.plan_s * 1e3, A.info.upload_s * 1e3, (t2 - t1) * 1e3, B.info.col_panels, B.info.plan_s * 1e3, B.info.upload_s * 1e3, same,
        A.bench(3, 20) * 1e6, B.bench(3, 20) * 1e6), flush=True)
    A.close(); B.close()

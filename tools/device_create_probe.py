import time, torch, numpy as np, cvr_amd
from cvr_amd import synth
n, nc, rp, ci, va = synth.web_google_like(1.0)[:5]
dev = torch.device("cuda", 0)
trp, tci, tva = torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); A = cvr_amd.CvrMatrix(n, nc, rp, ci, va); t1 = time.perf_counter()
    B = cvr_amd.CvrMatrix.from_device(n, nc, trp.data_ptr(), tci.data_ptr(), tva.data_ptr()); t2 = time.perf_counter()
    print("host arrays: create+preprocess %.2f ms (plan %.2f upload %.2f convert %.3f) | device arrays: %.2f ms (plan %.2f upload %.2f convert %.3f)" % (
        (t1 - t0) * 1e3, A.info.plan_s * 1e3, A.info.upload_s * 1e3, A.info.convert_s * 1e3, (t2 - t1) * 1e3, B.info.plan_s * 1e3, B.info.upload_s * 1e3, B.info.convert_s * 1e3))
    A.close(); B.close()

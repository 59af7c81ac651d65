"""independent SpMVs (same A, same or different x) alternating on two HIP streams with their own y buffers: throughput when
the fixed per-launch part of one SpMV overlaps the body of the other (PYTHONPATH=. python tools/two_stream_probe.py)"""
import time
import torch
import cvr_amd
from cvr_amd import synth

nrows, ncols, rp, ci, va = synth.web_google_like(1.0)[:5]
A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
dev = torch.device("cuda", 0)
x = torch.zeros(A.info.x_elems, dtype=torch.float64, device=dev)
x[:ncols] = torch.from_numpy(synth.x_rand(ncols)).to(dev)
ys = [torch.zeros(A.info.yext_elems, dtype=torch.float64, device=dev) for _ in range(4)]
st = [torch.cuda.Stream(device=dev) for _ in range(4)]
n = 2000
for nstreams in (1, 2, 3, 4):
    for _ in range(100):
        for s in range(nstreams):
            A.spmv_device(x.data_ptr(), ys[s].data_ptr(), st[s].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        s = i % nstreams
        A.spmv_device(x.data_ptr(), ys[s].data_ptr(), st[s].cuda_stream)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    print("%d stream(s): %.2f us per SpMV, %.1f GFLOP/s; results equal: %s" % (nstreams, t * 1e6, 2 * len(ci) / t / 1e9, bool(torch.equal(ys[0][:nrows], ys[1][:nrows])) if nstreams >= 2 else "-"))

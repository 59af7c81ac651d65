"""host enqueue cost vs GPU time of the sharded SpMV + all-gather loop with a 1-rank RCCL communicator (PYTHONPATH=. python tools/diag_gather.py)"""
import time, torch, numpy as np, cvr_amd
from cvr_amd import synth
nrows, ncols, rp, ci, va = synth.web_google_like(1.0)[:5]
A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
comm = cvr_amd.Comm(cvr_amd.comm_unique_id(), 1, 0, 0)
dev = torch.device("cuda", 0)
x = torch.zeros(A.info.x_elems, dtype=torch.float64, device=dev); x[:ncols] = torch.from_numpy(synth.x_rand(ncols)).to(dev)
ny = max(A.info.yext_elems, nrows)
ys = [torch.zeros(ny, dtype=torch.float64, device=dev) for _ in range(2)]
yalls = [torch.zeros(nrows, dtype=torch.float64, device=dev) for _ in range(2)]
st = torch.cuda.Stream(device=dev)
def run(n, ov=True):
    t0 = time.perf_counter()
    A.spmv_gather(comm, x.data_ptr(), [t.data_ptr() for t in ys], [t.data_ptr() for t in yalls], nrows, n, st.cuda_stream, overlap=ov)
    t1 = time.perf_counter()
    st.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
run(100)
print("library loop, overlapped: enqueue us/step %.1f total us/step %.1f" % run(2000))
run(100, False)
print("library loop, in order: enqueue us/step %.1f total us/step %.1f" % run(2000, False))
def run2(n):
    t0 = time.perf_counter()
    for k in range(n):
        A.spmv_device(x.data_ptr(), ys[0].data_ptr(), st.cuda_stream)
        comm.all_gather(ys[0].data_ptr(), yalls[0].data_ptr(), nrows, False, st.cuda_stream)
    t1 = time.perf_counter(); st.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
run2(100)
print("in-stream python loop: enqueue us/step %.1f total %.1f" % run2(2000))
def run3(n):
    t0 = time.perf_counter()
    for k in range(n):
        comm.all_gather(ys[0].data_ptr(), yalls[0].data_ptr(), nrows, False, st.cuda_stream)
    t1 = time.perf_counter(); st.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
run3(100)
print("all_gather only: enqueue us %.1f total %.1f" % run3(2000))
def run4(n):
    t0 = time.perf_counter()
    A.spmv_device(x.data_ptr(), ys[0].data_ptr(), st.cuda_stream, repeat=n)
    t1 = time.perf_counter(); st.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
run4(100)
print("spmv only: enqueue us %.1f total %.1f" % run4(2000))

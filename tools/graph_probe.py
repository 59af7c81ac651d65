"""per-step time of back-to-back SpMV launches against a captured hipGraph replay, for the whole web-Google-shaped matrix
and for one of 8 row shards (PYTHONPATH=. python tools/graph_probe.py)"""
import time
import torch
import cvr_amd
from cvr_amd import shard, synth

nrows, ncols, rp, ci, va = synth.web_google_like(1.0)[:5]
dev = torch.device("cuda", 0)
for nparts in (1, 8):
    b = shard.row_partition(rp, nparts)
    lrows, lrp, lci, lva = shard.local_csr(rp, ci, va, b, 0)
    A = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, tune_steps=nparts > 1)
    x = torch.zeros(A.info.x_elems, dtype=torch.float64, device=dev)
    x[:ncols] = torch.from_numpy(synth.x_rand(ncols)).to(dev)
    y = torch.zeros(A.info.yext_elems, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    n = 1000
    with torch.cuda.stream(st):
        A.spmv_device(x.data_ptr(), y.data_ptr(), st.cuda_stream, repeat=50)
        st.synchronize()
        t0 = time.perf_counter()
        A.spmv_device(x.data_ptr(), y.data_ptr(), st.cuda_stream, repeat=n)
        st.synchronize()
        t_stream = (time.perf_counter() - t0) / n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        A.spmv_device(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream, repeat=100)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 100):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / n
    print("1/%d of the matrix (S = %d): %.2f us per SpMV with stream launches, %.2f us in a replayed graph of 100" % (nparts, A.info.steps_per_chunk, t_stream * 1e6, t_graph * 1e6), flush=True)
    A.close()

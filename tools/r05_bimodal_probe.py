#!/usr/bin/env python3
"""tools/r05_bimodal_probe.py -- the stream-bound shapes (banded) ran 178 or 210 us from process to process: is it the placement of the image's
allocation?  One process, the banded 3.5 M matrix built once on the device, the handle created, timed and destroyed several times; a 1-GiB copy
kernel between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cvr_amd
from cvr_amd import synth_dev as D, capi
n = 3_500_000
bounds, nnz = D.banded_partition(n, 13, 1, 0)
rp, ci, va = D.banded_rows(n, 0, n, device="cuda")
torch.cuda.synchronize()
x = torch.zeros(n + 1, dtype=torch.float64, device="cuda"); x[:n] = torch.rand(n, dtype=torch.float64, device="cuda")
for rep in range(6):
    A = cvr_amd.CvrMatrix.from_device(n, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), is_f32=False)
    y = torch.zeros(max(A.info.yext_elems, 1), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for k in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        A.spmv_device(x.data_ptr(), y.data_ptr(), s, repeat=20)
        a.record(); A.spmv_device(x.data_ptr(), y.data_ptr(), s, repeat=100); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 10)
    gbs = capi.device_copy_bench(0, 1 << 30, 10) if hasattr(capi, "device_copy_bench") else float("nan")
    print(f"handle {rep}: SpMV {ts[0]:.1f} / {ts[1]:.1f} / {ts[2]:.1f} us, copy kernel {gbs:.0f} GB/s, image {A.info.image_bytes / 1e6:.0f} MB", flush=True)
    A.close()
    if rep == 2:          # (some allocations of other sizes in between)
        junk = [torch.empty(int(37e6) * (i + 1), dtype=torch.uint8, device="cuda") for i in range(5)]
        del junk
        torch.cuda.empty_cache()

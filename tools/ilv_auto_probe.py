"""tools/ilv_auto_probe.py -- the automatic rule for interleaved chunks against interleave = 0 on the shapes that get column panels:
us per SpMV (events over 50 launches) and wrong rows, per shape."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import cvr_amd
from cvr_amd import synth, synth_dev as D

dev = torch.device("cuda", 0)


def shapes(which):
    if "livejournal" in which:
        n, nc, rp, ci, va = synth.livejournal_like()
        yield "livejournal", n, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev)
    if "orkut" in which:
        n, rp, ci, va = D.orkut_like(device=dev)
        yield "orkut", n, rp, ci, va
    if "wikitalk" in which:
        n, rp, ci, va = D.wikitalk_like(device=dev)
        yield "wikitalk", n, rp, ci, va
    if "tiny_rows" in which:
        n = 2_400_000
        g = torch.Generator(device=dev); g.manual_seed(3)
        rp = torch.arange(n + 1, dtype=torch.int64, device=dev)
        near = torch.rand(n, generator=g, device=dev) < 0.5
        ci = torch.where(near, (torch.arange(n, device=dev) + torch.randint(-200, 200, (n,), generator=g, device=dev)).clamp(0, n - 1), torch.randint(0, n, (n,), generator=g, device=dev)).to(torch.int32)
        yield "tiny_rows", n, rp, ci, torch.randn(n, generator=g, device=dev, dtype=torch.float64)
    if "webgoogle3" in which:          # a web-Google shape three times the size (x = 22 MB: the mid range that gets eight panels)
        n, nc, rp, ci, va = synth.power_law_graph(int(916_428 * 3), int(5_105_039 * 3), 0.193, 456, 20261002)
        yield "webgoogle x3", n, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev)
    for f in (2.2, 2.6):
        if f"webgoogle{f}" in which:
            n, nc, rp, ci, va = synth.power_law_graph(int(916_428 * f), int(5_105_039 * f), 0.193, 456, 20261002)
            yield f"webgoogle x{f}", n, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev)
    for s in (24,):
        if f"rmat{s}" in which:
            rp, ci, va = D.rmat_rows(s, 0, 1 << s, device=dev)
            yield f"rmat{s}", 1 << s, rp, ci, va


which = sys.argv[1:] or ["orkut", "wikitalk", "tiny_rows", "webgoogle3", "webgoogle2.2", "webgoogle2.6"]
for name, n, rp, ci, va in shapes(which):
    f32 = va.dtype == torch.float32
    tdt = va.dtype
    x_full = D.x_rand(n, device=dev, dtype=tdt)
    yref, absy = D.csr_spmv_reference(rp, ci, va, x_full)
    for ilv in (0, -1, 1):
        A = cvr_amd.CvrMatrix.from_device(n, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), is_f32=f32, interleave=ilv)
        i = A.info
        x = torch.zeros(i.x_elems, dtype=tdt, device=dev); x[:n] = x_full
        y = torch.zeros(max(i.yext_elems, 1), dtype=tdt, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        A.spmv_device(x.data_ptr(), y.data_ptr(), s, repeat=5)
        a.record(); A.spmv_device(x.data_ptr(), y.data_ptr(), s, repeat=50); b.record(); torch.cuda.synchronize()
        wrong = int(torch.count_nonzero((y[:n].to(torch.float64) - yref).abs() > (1e-5 if f32 else 1e-12) * absy + 1e-300).item())
        print(f"{name:14s} interleave {ilv:2d} -> ilv {i.interleave} panels {i.col_panels:2d} S {i.steps_per_chunk:3d} wpb {i.waves_per_block} chunks {i.nchunks:6d} hub {i.hub_entries:5d} "
              f"image {i.image_bytes / 1e6:7.1f} MB  {a.elapsed_time(b) * 1e3 / 50:8.1f} us  wrong {wrong}  t_pre {1e3 * (i.plan_s + i.probe_s + i.dict_s + i.preprocess_wall_s):.2f} ms", flush=True)
        A.close()
        del x, y

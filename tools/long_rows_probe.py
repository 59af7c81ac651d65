"""tools/long_rows_probe.py -- what a split by row length buys on R-MAT: the rows of at least L non-zeros through the long-row prototype
(tools/ubench/long_rows), the rest (those rows emptied) through the library with its own rules.  python tools/long_rows_probe.py rmat22 L ..."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cvr_amd
from cvr_amd import synth_dev as D
import sorted_probe as SP

which = sys.argv[1]
Ls = [int(v) for v in sys.argv[2:]] or [256, 512, 1024]
dev = torch.device("cuda", 0)
scale = int(which[4:])
n = 1 << scale
rp, ci, va = D.rmat_rows(scale, 0, n, device=dev)
path = f"/tmp/{which}.csr"
SP.write_csr(path, n, n, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy(), True)
deg = rp[1:] - rp[:-1]
x_full = D.x_rand(n, device=dev, dtype=torch.float32)


def time_lib(rp_s, ci_s, va_s, label):
    A = cvr_amd.CvrMatrix.from_device(n, n, rp_s.data_ptr(), ci_s.data_ptr(), va_s.data_ptr(), is_f32=True)
    i = A.info
    x = torch.zeros(i.x_elems, dtype=torch.float32, device=dev); x[:n] = x_full
    y = torch.zeros(max(i.yext_elems, 1), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    A.spmv_device(x.data_ptr(), y.data_ptr(), s, repeat=5)
    a.record(); A.spmv_device(x.data_ptr(), y.data_ptr(), s, repeat=50); b.record(); torch.cuda.synchronize()
    yref, absy = D.csr_spmv_reference(rp_s, ci_s, va_s, x_full)
    wrong = int(torch.count_nonzero((y[:n].to(torch.float64) - yref).abs() > 1e-5 * absy + 1e-300).item())
    print(f"{label}: nnz {int(rp_s[-1])} panels {i.col_panels} S {i.steps_per_chunk} wpb {i.waves_per_block} hub {i.hub_entries} (share {i.hub_share:.2f}) ilv {i.interleave} -> {a.elapsed_time(b) * 1e3 / 50:.1f} us wrong {wrong}", flush=True)
    A.close()


time_lib(rp, ci, va, f"{which} whole")
for L in Ls:
    keep = torch.repeat_interleave(deg < L, deg)
    d2 = torch.where(deg < L, deg, torch.zeros_like(deg))
    rp_s = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp_s[1:] = torch.cumsum(d2, 0)
    time_lib(rp_s, ci[keep].contiguous(), va[keep].contiguous(), f"{which} rows < {L}")
    for gmax in (8, 32):
        r = subprocess.run([os.path.join(ROOT, "tools", "ubench", "long_rows"), path, "f32", str(L), str(gmax), "50"], capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-500:], flush=True)

"""tools/ilv_pre_probe.py lj|orkut -- preprocessing clocks and SpMV time of the interleaved layout on a large stand-in (three builds in one process:
the first pays for code loading)."""
import sys, os, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import cvr_amd
from cvr_amd import synth, synth_dev as D
which = sys.argv[1]
if which == 'lj':
    n, nc, rp, ci, va = synth.livejournal_like()
else:
    n, rp_t, ci_t, va_t = D.orkut_like(device='cuda'); nc = n
    rp, ci, va = rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy(); del rp_t, ci_t, va_t
x = synth.x_rand(nc)
yref = cvr_amd.csr_spmv_host(rp, ci, va, x, nthreads=16)
for rep in range(3):
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
    y, _ = A.spmv(x)
    i = A.info
    t = A.bench(5, 30)
    print(which, os.environ.get('CVR_DEBUG'), f"S {i.steps_per_chunk} ilv {i.interleave} panels {i.col_panels} pre_wall {i.preprocess_wall_s*1e3:.2f} ms convert {i.convert_s*1e3:.2f} plan {i.plan_s*1e3:.2f} probe {i.probe_s*1e3:.2f} hub {i.hub_select_s*1e3:.2f} dict {i.dict_s*1e3:.2f} spmv {t*1e6:.1f} us maxerr {np.max(np.abs(y-yref)):.2e}", flush=True)
    A.close()

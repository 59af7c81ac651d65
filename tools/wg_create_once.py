"""creates the web-Google-shaped matrix twice from host arrays (second create: warm) and prints the preprocessing figures of cvr_info
(arguments: nodict = value_dict off, f32 = random fp32 values)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth
n, nc, rp, ci, va = synth.web_google_like(1.0)[:5]
kw = {}
if "nodict" in sys.argv: kw["value_dict"] = 0
if "f32" in sys.argv:
    import numpy as np
    va = np.random.default_rng(1).standard_normal(len(ci)).astype(np.float32)
A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw); A.close()      # first use: runtime start-up
for rep in range(3):
    t0 = time.perf_counter()
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
    i = A.info
    print("create + preprocess %.0f us: plan %.0f probe %.0f hub %.0f dict %.0f convert(dev) %.0f preprocess wall %.0f upload %.0f  => T_pre %.0f us" % (
        (time.perf_counter() - t0) * 1e6, i.plan_s * 1e6, i.probe_s * 1e6, i.hub_select_s * 1e6, i.dict_s * 1e6, i.convert_s * 1e6, i.preprocess_wall_s * 1e6, i.upload_s * 1e6,
        (i.plan_s + i.probe_s + i.hub_select_s + i.dict_s + i.preprocess_wall_s) * 1e6))
    A.close()

import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np, cvr_amd
from cvr_amd import synth, capi
n,nc,rp,ci,va = synth.web_google_like()
import os
A = cvr_amd.CvrMatrix(n,nc,rp,ci,va, steps_per_chunk=48, waves_per_block=7, x_window=8192, col_phases=12, xcd_swizzle=int(os.environ.get('SWZ','1')))
x = synth.x_rand(nc)
A.spmv(x)
A.bench(20, 50)
L = capi.lib()
buf = np.zeros(8*4096, dtype=np.uint64)
L.cvr_debug_timeline.argtypes=[C.c_void_p, C.c_longlong]
print("rc", L.cvr_debug_timeline(buf.ctypes.data, len(buf)))
t = buf.reshape(-1,8)[:A.info.nchunks,:6].astype(np.int64)
t0 = t[:,0].min()
t = (t - t0) / 100.0   # wall_clock64 = 100 MHz -> us
print("chunks", len(t))
for name, col in (("start",0),("begin_chunk",4),("loads issued",5),("after barrier",1),("after loop",2),("end",3)):
    v = t[:,col]; print(f"{name:14s} min {v.min():6.2f} p10 {np.percentile(v,10):6.2f} median {np.median(v):6.2f} p90 {np.percentile(v,90):6.2f} max {v.max():6.2f}")
d = t[:,2]-t[:,1]; print("loop duration  median %.2f p10 %.2f p90 %.2f max %.2f" % (np.median(d), np.percentile(d,10), np.percentile(d,90), d.max()))
d = t[:,1]-t[:,0]; print("prologue       median %.2f p90 %.2f max %.2f" % (np.median(d), np.percentile(d,90), d.max()))
d = t[:,3]-t[:,2]; print("epilogue       median %.2f p90 %.2f max %.2f" % (np.median(d), np.percentile(d,90), d.max()))

"""creates the LiveJournal-shaped matrix once from host arrays (for `rocprofv3 --hip-trace --stats -- python3 tools/lj_create_once.py`)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvr_amd
from cvr_amd import synth
n, nc, rp, ci, va = synth.livejournal_like(1.0)[:5]
A = cvr_amd.CvrMatrix(n, nc, rp, ci, va); A.close()      # first use: runtime start-up
t0 = time.perf_counter()
A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
print("create + preprocess %.0f ms: plan %.0f, upload %.0f, convert %.1f" % ((time.perf_counter() - t0) * 1e3, A.info.plan_s * 1e3, A.info.upload_s * 1e3, A.info.convert_s * 1e3))
A.close()

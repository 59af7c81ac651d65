"""round 5 diagnostic: the y that bench.py --gpus 8 (one device, gloo) dumps, against the oracle on the matrix built in one piece, slice by slice"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
if os.environ.get("DEBUG_TORCH_FIRST"):
    import torch          # (as under pytest: conftest.py imports torch before anything runs)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, CVR_BENCH_ONE_DEVICE="1", CVR_BENCH_NO_TUNE="1", CVR_BENCH_DEBUG_DUMP="1")
for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "2", "--workload", "rmat20", "--no-cpu-baseline", "--dump-y", "/tmp/y8.npy"],
                   capture_output=True, text=True, timeout=900, env=env)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
print({k: d.get(k) for k in ("verdict_wrong_rows", "gathered_slices_differing_between_ranks", "gather_impl")})
import torch
from cvr_amd import synth, synth_dev as D
import oraclelib as O
n = 1 << 20
rp, ci, va = [t.cpu().numpy() for t in D.rmat_rows(20, 0, n, device="cuda")]
x = synth.x_rand(n, np.float32)
yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
y = np.load("/tmp/y8.npy"); raw = np.load("/tmp/y8.npy.raw.npy"); pick = np.load("/tmp/y8.npy.pick.npy"); b = np.load("/tmp/y8.npy.bounds.npy")
print("bounds", b.tolist(), "max_rows", len(raw) // 8, "y", y.shape, y.dtype, "raw", raw.shape)
print("dump == raw[pick]:", bool(np.array_equal(y, raw[pick])))
mr = len(raw) // 8
for p in range(8):
    lo, hi = int(b[p]), int(b[p + 1])
    sl = raw[p * mr: p * mr + hi - lo].astype(np.float64)
    bad = np.abs(sl - yref[lo:hi]) > 1e-5 * absy[lo:hi] + 1e-300
    bad_dump = np.abs(y[lo:hi].astype(np.float64) - yref[lo:hi]) > 1e-5 * absy[lo:hi] + 1e-300
    print(f"slice {p}: rows [{lo},{hi}) raw wrong {int(bad.sum())}, dump wrong {int(bad_dump.sum())}, raw zeros {int((sl == 0).sum())} of {hi - lo}, nonzero refs {int((yref[lo:hi] != 0).sum())}")
# the same matrix through one handle in this process
import cvr_amd
A = cvr_amd.CvrMatrix(n, n, rp, ci, va)
y1, _ = A.spmv(x)
print("one handle vs oracle wrong:", int((np.abs(y1.astype(np.float64) - yref) > 1e-5 * absy + 1e-300).sum()), "; dump vs one handle wrong:", int((np.abs(y.astype(np.float64) - y1.astype(np.float64)) > 2e-5 * absy + 1e-300).sum()))
# every rank's own view (CVR_BENCH_DEBUG_DUMP): shard arrays, own y, own reference, own copy of the gathered vector
import hashlib
print("x as the test builds it == rank 0's x:", bool(np.array_equal(x, np.load("/tmp/y8.npy.rank0.npz")["x"])))
for p in range(8):
    z = np.load(f"/tmp/y8.npy.rank{p}.npz")
    lo, hi = int(b[p]), int(b[p + 1]); e0, e1 = int(rp[lo]), int(rp[hi])
    want = [hashlib.sha1(a.tobytes()).hexdigest() for a in ((rp[lo:hi + 1] - rp[lo]).astype(rp.dtype), ci[e0:e1], va[e0:e1])]
    L = int(z["last"][0])
    def nbad(v, ref=yref[lo:hi], ab=absy[lo:hi]):
        return int((np.abs(v.astype(np.float64) - ref) > 1e-5 * ab + 1e-300).sum())
    own = [z["yall0"], z["yall1"]][L]
    per = [nbad(own[q * mr: q * mr + int(b[q + 1] - b[q])], yref[int(b[q]):int(b[q + 1])], absy[int(b[q]):int(b[q + 1])]) for q in range(8)]
    print(f"rank {p}: shard arrays equal the slice {[a == w for a, w in zip(z['hashes'].tolist(), want)]}; own y0 wrong {nbad(z['y0'])}, own y1 wrong {nbad(z['y1'])}; "
          f"own reference vs oracle wrong {nbad(z['yref'])}; x equal {bool(np.array_equal(z['x'], x))}; last {L}; its gathered copy, wrong rows per slice {per}")

"""tools/sorted_probe.py -- drives tools/ubench/sorted_spmv (the interleaved-chunk prototype, round 4) on the stand-in matrices of
bench.py: writes the CSR to /tmp as a raw file and runs the binary over a list of configurations.
  python tools/sorted_probe.py lj|rmat22|webgoogle|orkut|wikitalk "R W Smax P mode depth [dict [noadd]]" ...
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_csr(path, nrows, ncols, rp, ci, va, f32):
    with open(path, "wb") as f:
        np.array([nrows, ncols, len(ci)], dtype=np.int64).tofile(f)
        np.asarray(rp, dtype=np.int64).tofile(f)
        np.asarray(ci, dtype=np.int32).tofile(f)
        np.asarray(va, dtype=np.float32 if f32 else np.float64).tofile(f)


def main():
    which = sys.argv[1]
    cfgs = sys.argv[2:]
    from cvr_amd import synth
    t0 = time.time()
    f32 = False
    if which == "lj":
        n, nc, rp, ci, va = synth.livejournal_like()
    elif which == "webgoogle":
        n, nc, rp, ci, va = synth.web_google_like()
    elif which.startswith("rmat"):
        import torch
        from cvr_amd import synth_dev as D
        scale = int(which[4:] or 22)
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        rp_t, ci_t, va_t = D.rmat_rows(scale, 0, 1 << scale, device=dev)
        n = nc = 1 << scale
        rp, ci, va = rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy()
        del rp_t, ci_t, va_t
        f32 = True
    elif which in ("orkut", "wikitalk"):
        import torch
        from cvr_amd import synth_dev as D
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        sc = float(os.environ.get("PROBE_SCALE", "1.0"))
        n, rp_t, ci_t, va_t = (D.orkut_like if which == "orkut" else D.wikitalk_like)(scale=sc, device=dev)
        nc = n
        rp, ci, va = rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy()
        del rp_t, ci_t, va_t
        torch.cuda.empty_cache() if dev == "cuda" else None
    else:
        fn = getattr(synth, which)
        n, nc, rp, ci, va = fn()
        f32 = va.dtype == np.float32
    path = f"/tmp/{which}.csr"
    write_csr(path, n, nc, rp, ci, va, f32)
    print(f"# {which}: {n} x {nc}, nnz {len(ci)}, built + written in {time.time() - t0:.1f} s", flush=True)
    for c in cfgs:
        a = c.split()
        env = dict(os.environ)
        while a and "=" in a[0]:                       # leading NAME=value tokens: the run's environment (TOK_U=2, EXE=sorted_spmv_nt, SAME_STREAM=8 ...)
            k, v = a.pop(0).split("=", 1)
            env[k] = v
        exe = os.path.join(ROOT, "tools", "ubench", env.get("EXE", "sorted_spmv"))
        R, W, S, P, mode, depth = a[:6]
        rest = a[6:]
        cmd = [exe, path, "f32" if f32 else "f64", R, W, S, P, mode, depth, "50"] + rest
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        print(f"## {c}\n{r.stdout}{r.stderr[-2000:] if r.returncode else ''}", flush=True)


if __name__ == "__main__":
    main()

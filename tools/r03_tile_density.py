"""tools/r03_tile_density.py -- what a 2-D form with MOVING x tiles in LDS could take off the L2s (CPU only, numpy): rows cut into blocks of R rows
(their sums in LDS), columns into tiles of W values (staged into LDS with coalesced loads: W * vs / 128 requests); a (block, tile) pair is
worth staging when it holds at least `gain` times as many non-zeros as the staging costs requests.  Prints, per (R, W), the share of the
non-zeros in such pairs and the L1->L2 requests that would be left (gathers outside + staging), against one request per non-zero now.
(python tools/r03_tile_density.py livejournal|webgoogle|rmat22)"""
import sys
import numpy as np
from cvr_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "livejournal"
if name == "livejournal":
    n, nc, rp, ci, va = synth.livejournal_like()[:5]
elif name.startswith("rmat"):
    n, nc, rp, ci, va = synth.rmat(int(name[4:]), dtype=np.float32)[:5]
else:
    n, nc, rp, ci, va = synth.web_google_like()[:5]
vs = va.dtype.itemsize
nnz = len(ci)
rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
print(f"# {name}: {n} x {nc}, nnz {nnz}, x {nc * vs / 1e6:.1f} MB")
for R in (4096, 8192, 16384):
    for W in (4096, 8192):
        ntile = (nc + W - 1) // W
        key = (rows // R) * ntile + ci.astype(np.int64) // W
        cnt = np.bincount(np.unique(key, return_inverse=True)[1])
        stage = W * vs // 128
        line = []
        for gain in (2, 4):
            good = cnt >= gain * stage
            inside = int(cnt[good].sum())
            left = nnz - inside + int(good.sum()) * stage
            line.append(f"gain>={gain}: {inside / nnz * 100:5.1f} % of nnz in {int(good.sum())} pairs ({good.sum() / (n / R):.1f} per block), requests left {left / 1e6:.1f} M = {left / nnz * 100:.0f} %")
        lds = R * vs + 2 * W * vs
        print(f"R {R:6d} W {W:5d} (LDS {lds // 1024} KiB with two tile buffers): " + " | ".join(line), flush=True)

#!/usr/bin/env python3
"""tools/make_mtx.py OUT.mtx [webgoogle|livejournal] [scale] -- write a seeded synthetic matrix as a row-major `pattern general`
Matrix-Market file (1-based coordinates): input for ./spmv.cvr and for the reference binary."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvr_amd import synth

out = sys.argv[1]
kind = sys.argv[2] if len(sys.argv) > 2 else "webgoogle"
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
n, nc, rp, ci, va = (synth.web_google_like if kind == "webgoogle" else synth.livejournal_like)(scale=scale)
rows = np.repeat(np.arange(1, n + 1, dtype=np.int64), np.diff(rp))
with open(out, "w") as f:
    f.write("%%MatrixMarket matrix coordinate pattern general\n")
    f.write(f"{n} {nc} {len(ci)}\n")
    step = 1 << 22
    for a in range(0, len(ci), step):          # in pieces: the text of 69 M entries at once is gigabytes
        np.savetxt(f, np.column_stack([rows[a:a + step], ci[a:a + step].astype(np.int64) + 1]), fmt="%d %d")
print(f"{out}: {n} x {nc}, {len(ci)} entries")

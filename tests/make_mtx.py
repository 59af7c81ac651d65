#!/usr/bin/env python3
"""tests/make_mtx.py OUT.mtx [webgoogle|livejournal] [scale] -- write a seeded synthetic matrix as a row-major
`pattern general` Matrix-Market file (input for ./spmv.cvr and for the reference binary).  Uses the oracle
library's writer (test/bench infrastructure)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as O
from cvr_amd import synth

out = sys.argv[1]
kind = sys.argv[2] if len(sys.argv) > 2 else "webgoogle"
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
n, nc, rp, ci, va = (synth.web_google_like if kind == "webgoogle" else synth.livejournal_like)(scale=scale)
O.write_mtx_pattern(out, n, nc, rp, ci)
print(f"{out}: {n} x {nc}, {len(ci)} entries")

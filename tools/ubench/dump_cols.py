"""writes the column indices of the web-Google-shaped matrix (CSR order) as raw u32 for tools/ubench/scalar_gather"""
import sys
import numpy as np
from cvr_amd import synth
n, nc, rp, ci, va = synth.web_google_like(1.0)[:5]
ci.astype(np.uint32).tofile(sys.argv[1])
print(len(ci))

#!/usr/bin/env python3
"""tools/r03_rmat_reorder_table_sweep.py [scale] -- R-MAT fp64 with x re-ordered by popularity: how large the hub table beside it should be
(the table alone loses on fp64: DESIGN 5.3; the re-ordering is what wins).  PYTHONPATH=. python tools/r03_rmat_reorder_table_sweep.py 22"""
import sys
import torch
import cvr_amd
from cvr_amd import synth_dev as D

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dev = torch.device("cuda", 0)
n = 1 << scale
lrp, lci, lva = D.rmat_rows(scale, 0, n, device=dev)
lva = lva.double()
torch.cuda.synchronize()
for table, reorder in ((-1, -1), (0, 0), (1024, 1), (2048, 1), (4096, 1), (8192, 1), (12288, 1), (-1, 1)):
    A = cvr_amd.CvrMatrix.from_device(n, n, lrp.data_ptr(), lci.data_ptr(), lva.data_ptr(), is_f32=False, hub_table=table, hub_reorder=reorder)
    i = A.info
    t = A.bench(5, 50)
    print(f"R-MAT-{scale} fp64 hub_table {table} hub_reorder {reorder}: entries {i.hub_entries} reorder {i.hub_reorder} w {i.waves_per_block} S {i.steps_per_chunk} lds {i.lds_bytes}: {t * 1e6:8.1f} us", flush=True)
    A.close()

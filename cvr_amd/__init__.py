"""cvr_amd -- MI355X-native CVR-format SpMV (the hot path of puckbee/CVR, /root/reference/spmv.cpp).

The product is the C-ABI library `libcvr_amd.so` (include/cvr_amd.h: hand-written gfx950 kernels behind an
opaque handle) and the host program `spmv.cvr` with the reference's CLI.  This package is the thin
Python view of that ABI used by tests/, bench.py and __graft_entry__.py; it contains no compute of its
own and there is no CPU fallback: without the library or without a GPU the calls raise.
"""
from .capi import (Comm, CvrError, CvrMatrix, MultiMatrix, row_partition, comm_unique_id, device_count, last_error, lib, lib_path, load_mm, plan_chunks, plan_selfcheck,  # noqa: F401
                   csr_spmv_host, fill_x, verdict, version)

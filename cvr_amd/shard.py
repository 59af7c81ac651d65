"""Row sharding of one SpMV over the GPUs of a node: contiguous row blocks with balanced nnz, cut at row
boundaries (the reference balances nnz per thread the same way, spmv.cpp:584-667, but cuts inside rows and
repairs with atomics); x replicated; y slices all-gathered (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import numpy as np


def row_partition(row_ptr, nparts, row_cost_milli=0):
    """bounds[nparts+1]: part p owns rows bounds[p] .. bounds[p+1]-1; cost per part (non-zeros + row_cost_milli / 1000 per row) as
    equal as row boundaries allow; row_cost_milli = 0: non-zeros alone, the reference's rule (spmv.cpp:584-627)"""
    from . import capi
    return capi.row_partition(row_ptr, nparts, row_cost_milli)      # the library's one partition rule (cvr_row_partition_cost, cvr_multi.hip)


def local_csr(row_ptr, col_idx, vals, bounds, p):
    """the rebased CSR arrays of part p (views / small copies)"""
    rp = np.asarray(row_ptr, dtype=np.int64)
    b, e = int(bounds[p]), int(bounds[p + 1])
    lo, hi = int(rp[b]), int(rp[e])
    return e - b, rp[b:e + 1] - lo, col_idx[lo:hi], vals[lo:hi]


def gather_layout(bounds):
    """equal-count all-gather: every rank contributes max_rows values; returns (max_rows, index array that
    picks the real rows out of the [nparts * max_rows] gathered buffer)"""
    sizes = np.diff(bounds)
    max_rows = int(sizes.max()) if len(sizes) else 0
    idx = np.concatenate([p * max_rows + np.arange(int(sizes[p]), dtype=np.int64) for p in range(len(sizes))]) \
        if len(sizes) else np.zeros(0, dtype=np.int64)
    return max_rows, idx


def all_gather_tensor(out, src):
    """dist.all_gather_into_tensor(out, src) (bench.py: the owners' references travel like y)"""
    import torch.distributed as dist
    dist.all_gather_into_tensor(out, src)
    return out


def all_gather_y(y_local, max_rows, out=None):
    """the one exchange step of the sharded SpMV: every rank contributes the first max_rows values of its
    y buffer (its own rows, then don't-care), every rank receives all slices.  torch.distributed: RCCL over
    xGMI for device tensors ("nccl" backend), gloo for the CPU tests."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * max_rows, dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(out, y_local[:max_rows])
    return out


def pipelined_steps(spmv, y_bufs, yall_bufs, max_rows, steps):
    """`steps` sharded SpMVs of the fixed-x loop (the reference's Ntimes loop, spmv.cpp:1024): step k computes into
    y_bufs[k % 2] and starts the all-gather of that slice asynchronously; the gather of step k overlaps the compute of
    step k + 1 and is waited for before its buffers are reused (step k + 2).  Every step's y is fully gathered on
    every rank when this returns.  spmv(y_buf) must enqueue the local SpMV into y_buf on the current stream."""
    import torch.distributed as dist
    pending = [None, None]
    for k in range(steps):
        b = k & 1
        if pending[b] is not None:
            pending[b].wait()              # the gather that last read y_bufs[b] / wrote yall_bufs[b]
        spmv(y_bufs[b])
        pending[b] = dist.all_gather_into_tensor(yall_bufs[b], y_bufs[b][:max_rows], async_op=True)
    for w in pending:
        if w is not None:
            w.wait()
    return (steps - 1) & 1                 # index of the buffers holding the last step's result

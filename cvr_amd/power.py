"""Iterative caller of the hot path: power iteration x <- A x / ||A x|| with everything
device-resident.  The realistic consumer of a web-graph SpMV, and the one setting where the all-gather of y is on
the critical path: with the rows sharded over the ranks, every iteration ends with the all-gather of the y slices,
from which each rank rebuilds its replicated x.  PyTorch provides the vectors, the norm and torch.distributed;
the SpMV itself is cvr_spmv_device (HIP kernels behind the C ABI).  Square matrices only."""
def power_iteration(A, nrows_total, bounds=None, rank=0, iters=20, x0=None):
    """A: CvrMatrix of this rank's row block (all rows when bounds is None).  Returns (eigenvalue estimate,
    x as a torch tensor of nrows_total values, seconds per iteration)."""
    import time
    import torch
    from . import shard
    dev = torch.device("cuda", torch.cuda.current_device())
    info = A.info
    assert info.ncols == nrows_total, "power iteration needs a square matrix"
    dt = torch.float32 if A.f32 else torch.float64
    x = torch.zeros(info.x_elems, dtype=dt, device=dev)           # x_ext: the pad slot x[ncols] stays 0
    x[:nrows_total] = (torch.ones(nrows_total, dtype=dt, device=dev) if x0 is None else torch.as_tensor(x0, dtype=dt, device=dev))
    x[:nrows_total] /= torch.linalg.vector_norm(x[:nrows_total])
    world = 1 if bounds is None else len(bounds) - 1
    if world > 1:
        max_rows, pick = shard.gather_layout(bounds)
        pick = torch.from_numpy(pick).to(dev)
    else:
        max_rows = 0
    y = torch.zeros(max(info.yext_elems, max_rows), dtype=dt, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    lam = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        A.spmv_device(x.data_ptr(), y.data_ptr(), stream)
        yfull = shard.all_gather_y(y, max_rows)[pick] if world > 1 else y[:nrows_total]
        lam_t = torch.dot(x[:nrows_total], yfull)                  # Rayleigh quotient (x is normalised)
        x[:nrows_total] = yfull / torch.linalg.vector_norm(yfull)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / max(iters, 1)
    lam = float(lam_t.item()) if iters else 0.0
    return lam, x[:nrows_total], dt_s

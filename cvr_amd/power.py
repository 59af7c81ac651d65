"""Iterative caller of the hot path: power iteration x <- A x / ||A x|| with everything
device-resident.  The realistic consumer of a web-graph SpMV, and the one setting where the all-gather of y is on
the critical path: with the rows sharded over the ranks, every iteration ends with the all-gather of the y slices,
from which each rank rebuilds its replicated x.  The loop itself is native (cvr_power_iteration in the C ABI: SpMV,
RCCL all-gather, fixed-tree dot products and the normalisation, no host round trip per iteration); PyTorch only owns the
x tensor here.  Square matrices only."""


def power_iteration(A, nrows_total, bounds=None, comm=None, iters=20, x0=None):
    """A: CvrMatrix of this rank's row block (all rows when bounds is None; with bounds, comm is the cvr_amd.Comm of the
    ranks).  Returns (eigenvalue estimate, x as a torch tensor of nrows_total values, seconds per iteration)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    info = A.info
    assert info.ncols == nrows_total, "power iteration needs a square matrix"
    dt = torch.float32 if A.f32 else torch.float64
    x = torch.zeros(info.x_elems, dtype=dt, device=dev)           # x_ext: the pad slot x[ncols] stays 0
    x[:nrows_total] = (torch.ones(nrows_total, dtype=dt, device=dev) if x0 is None else torch.as_tensor(x0, dtype=dt, device=dev))
    stream = torch.cuda.current_stream()
    stream.synchronize()
    lam, sec = A.power_iteration(x.data_ptr(), iters, comm=comm, bounds=bounds, stream=stream.cuda_stream)
    return lam, x[:nrows_total], sec

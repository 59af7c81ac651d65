"""Shard-local construction of the large synthetic workloads (BASELINE.json configs[3], [4]) with torch, on whatever device
the caller names (the GPU in bench.py, the CPU in the gloo tests): every rank builds only its own row block, so that an
R-MAT scale-26 or a 28-M-row banded matrix on 8 GPUs reaches the timed loop in seconds instead of building the whole matrix
in numpy on every rank (SURVEY.md 8(d) synthetic inputs, 8(e) partitioning).

Definitions (independent of how the rows are sharded, so a shard equals the slice of the whole):
  * R-MAT (Graph500 parameters): edge e belongs to block e // 2^24; the random numbers of block b at recursion level l come
    from torch.Generator(device).manual_seed(seed * 1000003 + b * 64 + l), the values from level 63.  The stream of a torch
    generator differs between CPU and GPU, so the matrix is defined per device type.
  * banded symmetric pattern: row r holds columns r - hb .. r + hb inside the matrix; value = splitmix64(r * 64 + k).
"""
import numpy as np
import torch

RMAT_BLOCK = 1 << 24


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    return g


def _rmat_block(scale, block, nedges, a, b, c, seed, device, want_cols=True):
    """rows (and columns) of the edges block * 2^24 ... of the R-MAT stream"""
    lo = block * RMAT_BLOCK
    m = min(RMAT_BLOCK, nedges - lo)
    rows = torch.zeros(m, dtype=torch.int64, device=device)
    cols = torch.zeros(m, dtype=torch.int64, device=device) if want_cols else None
    ab, abc = a + b, a + b + c
    for lvl in range(scale):
        r = torch.rand(m, generator=_gen(device, seed * 1000003 + block * 64 + lvl), device=device)
        rows |= (r >= ab).to(torch.int64) << lvl
        if want_cols:
            cols |= (((r >= a) & (r < ab)) | (r >= abc)).to(torch.int64) << lvl
    return rows, cols


def rmat_row_degrees(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=1, device="cpu"):
    n, m = 1 << scale, edge_factor << scale
    deg = torch.zeros(n, dtype=torch.int64, device=device)
    for blk in range((m + RMAT_BLOCK - 1) // RMAT_BLOCK):
        rows, _ = _rmat_block(scale, blk, m, a, b, c, seed, device, want_cols=False)
        deg += torch.bincount(rows, minlength=n)
    return deg


def partition_from_degrees(deg, nparts, row_cost_milli=0):
    """bounds[nparts + 1] with cost per part (non-zeros + row_cost_milli / 1000 per row) as equal as row boundaries allow
    (cvr_row_partition_cost on the prefix sums, restated on device tensors)"""
    rp = torch.zeros(len(deg) + 1, dtype=torch.int64, device=deg.device)
    rp[1:] = torch.cumsum(deg, 0)
    cost = rp * 1000 + torch.arange(len(deg) + 1, dtype=torch.int64, device=deg.device) * int(row_cost_milli)
    k = torch.arange(1, nparts, dtype=torch.int64, device=deg.device)
    targets = ((k * int(rp[-1])) // nparts) * 1000 + (k * (len(deg) * int(row_cost_milli))) // nparts
    cuts = torch.searchsorted(cost, targets, right=False)
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=deg.device), cuts.clamp(0, len(deg)),
                        torch.tensor([len(deg)], dtype=torch.int64, device=deg.device)])
    return torch.cummax(bounds, 0).values.cpu().numpy(), rp


def rmat_rows(scale, row_lo, row_hi, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=1, device="cpu", dtype=torch.float32):
    """CSR of rows [row_lo, row_hi) of the R-MAT matrix (duplicates kept, entries of a row sorted by column, ties in stream
    order): (row_ptr int64 [rows + 1] rebased to 0, col_idx int32, vals) as tensors on `device`"""
    n, m = 1 << scale, edge_factor << scale
    keep_r, keep_c, keep_v = [], [], []
    for blk in range((m + RMAT_BLOCK - 1) // RMAT_BLOCK):
        rows, cols = _rmat_block(scale, blk, m, a, b, c, seed, device)
        sel = (rows >= row_lo) & (rows < row_hi)
        v = torch.rand(len(rows), generator=_gen(device, seed * 1000003 + blk * 64 + 63), device=device, dtype=torch.float32)
        keep_r.append(rows[sel] - row_lo)
        keep_c.append(cols[sel])
        keep_v.append(v[sel].to(dtype))
    r, cc, v = torch.cat(keep_r), torch.cat(keep_c), torch.cat(keep_v)
    del keep_r, keep_c, keep_v
    order = torch.argsort(r * n + cc, stable=True)
    r, cc, v = r[order], cc[order].to(torch.int32), v[order]
    rp = torch.zeros(row_hi - row_lo + 1, dtype=torch.int64, device=device)
    rp[1:] = torch.cumsum(torch.bincount(r, minlength=row_hi - row_lo), 0)
    return rp, cc, v


def _splitmix_unit(z):
    """splitmix64 finaliser of int64 tensors -> float64 in [-1, 1)"""
    def mul(x, k):       # 64-bit wrap-around product with a constant given as unsigned
        return x * torch.tensor(k - (1 << 64) if k >= (1 << 63) else k, dtype=torch.int64, device=x.device)
    def shr(x, s):       # logical shift right on int64
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = mul(z + 1, 0x9E3779B97F4A7C15)
    z = mul(z ^ shr(z, 30), 0xBF58476D1CE4E5B9)
    z = mul(z ^ shr(z, 27), 0x94D049BB133111EB)
    z = z ^ shr(z, 31)
    return shr(z, 11).to(torch.float64) * (2.0 / 9007199254740992.0) - 1.0


def banded_partition(n, half_band, nparts, row_cost_milli=0):
    """nnz-balanced row bounds of the banded matrix, from the closed-form row lengths"""
    r = np.arange(n, dtype=np.int64)
    deg = np.minimum(r, half_band) + 1 + np.minimum(n - 1 - r, half_band)
    rp = np.concatenate([[0], np.cumsum(deg)])
    from . import shard
    return shard.row_partition(rp, nparts, row_cost_milli), int(rp[-1])


def banded_rows(n, row_lo, row_hi, half_band=13, device="cpu", dtype=torch.float64):
    """CSR of rows [row_lo, row_hi) of the n x n banded symmetric pattern (2 * half_band + 1 non-zeros per inner row)"""
    w = 2 * half_band + 1
    r = torch.arange(row_lo, row_hi, dtype=torch.int64, device=device)
    cols = r[:, None] + torch.arange(-half_band, half_band + 1, dtype=torch.int64, device=device)[None, :]
    ok = (cols >= 0) & (cols < n)
    k = torch.arange(w, dtype=torch.int64, device=device)[None, :].expand_as(cols)
    vals = _splitmix_unit((r[:, None] * 64 + k)[ok]).to(dtype)
    rp = torch.zeros(row_hi - row_lo + 1, dtype=torch.int64, device=device)
    rp[1:] = torch.cumsum(ok.sum(1), 0)
    return rp, cols[ok].to(torch.int32), vals


def x_rand(n, device="cpu", dtype=torch.float64):
    """the seeded x of cvr_amd.synth.x_rand (splitmix64(0xC0FFEE, j) -> [-1, 1)), built on the device"""
    j = torch.arange(n, dtype=torch.int64, device=device)
    def mul(x, k):
        return x * torch.tensor(k - (1 << 64) if k >= (1 << 63) else k, dtype=torch.int64, device=x.device)
    def shr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = 0xC0FFEE + mul(j + 1, 0x9E3779B97F4A7C15)
    z = mul(z ^ shr(z, 30), 0xBF58476D1CE4E5B9)
    z = mul(z ^ shr(z, 27), 0x94D049BB133111EB)
    z = z ^ shr(z, 31)
    return (shr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0).to(dtype)


def csr_spmv_reference(rp, ci, va, x):
    """fp64 y = A x and sum |a| |x| per row with torch (segment sums over the rows): the parity guard of bench.py for
    device-built shards"""
    nrows = len(rp) - 1
    y = torch.empty(nrows, dtype=torch.float64, device=rp.device)
    ay = torch.empty(nrows, dtype=torch.float64, device=rp.device)
    xd = x.to(torch.float64)
    step = 1 << 19                                  # rows per call (the segment kernel's grid is limited)
    for r0 in range(0, nrows, step):
        r1 = min(nrows, r0 + step)
        a, b = int(rp[r0]), int(rp[r1])
        prod = va[a:b].to(torch.float64) * xd[ci[a:b].to(torch.int64)]
        lengths = rp[r0 + 1:r1 + 1] - rp[r0:r1]
        y[r0:r1] = torch.segment_reduce(prod, "sum", lengths=lengths, unsafe=True)
        ay[r0:r1] = torch.segment_reduce(prod.abs(), "sum", lengths=lengths, unsafe=True)
    return y, ay


# ---- two more SuiteSparse-shaped power-law stand-ins (round 4): defined like R-MAT above -- by torch generator streams, so per device type --
# and built on the device in seconds.  Pattern matrices: value = (position in row-major order) % 13, what the reference's loader gives a
# `pattern` file (spmv.cpp:417), so the value dictionary applies as it does to the SNAP graphs themselves.
ORKUT = dict(n=3_072_441, edges=117_185_083, seed=20261004)          # com-Orkut: undirected friendship graph, mean degree 76, max 33 313
WIKITALK = dict(n=2_394_385, nnz=5_021_410, seed=20261005)           # wiki-Talk: 94 % of the users never write, a few write to 100 000 others


def _csr_from_keys(key, n, device):
    """sorted unique keys row * n + col -> (row_ptr int64, col_idx int32)"""
    r = torch.div(key, n, rounding_mode="floor")
    c = (key - r * n).to(torch.int32)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rp[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return rp, c


def _pattern_values(nnz, device, dtype=torch.float64):
    return (torch.arange(nnz, dtype=torch.int64, device=device) % 13).to(dtype)


def orkut_like(scale=1.0, device="cpu", dtype=torch.float64):
    """com-Orkut's shape: symmetric, ~3.07 M x 3.07 M, ~234 M non-zeros (mean 76 per row), degrees with a power-law tail, half of a user's
    friends inside the user's own community (ids nearby), half drawn by popularity.  Returns (n, row_ptr, col_idx, vals) on `device`."""
    n = max(1024, int(ORKUT["n"] * scale))
    m = max(4096, int(ORKUT["edges"] * scale))
    g = _gen(device, ORKUT["seed"])
    perm = torch.randperm(n, generator=g, device=device)
    keys = []
    step = 1 << 25
    for lo in range(0, m, step):                       # in pieces: the temporaries of 117 M edges at once are several GB
        k = min(step, m - lo)
        u = perm[(n * torch.rand(k, generator=g, device=device, dtype=torch.float64) ** 1.65).to(torch.int64).clamp_(0, n - 1)]
        pop = perm[(n * torch.rand(k, generator=g, device=device, dtype=torch.float64) ** 1.65).to(torch.int64).clamp_(0, n - 1)]
        e = torch.rand(k, generator=g, device=device, dtype=torch.float64)
        lap = (20000.0 * torch.sign(e - 0.5) * torch.log1p(-2.0 * (e - 0.5).abs().clamp_(max=0.4999999))).round().to(torch.int64)      # Laplace(0, 20 000)
        v = torch.where(torch.rand(k, generator=g, device=device) < 0.5, (u - lap).clamp_(0, n - 1), pop)
        keep = u != v
        a, b = torch.minimum(u, v)[keep], torch.maximum(u, v)[keep]
        keys.append(torch.unique(a * n + b))
        del u, pop, e, lap, v, keep, a, b
    und = torch.unique(torch.cat(keys))
    del keys
    a, b = torch.div(und, n, rounding_mode="floor"), und % n
    del und
    key = torch.sort(torch.cat([a * n + b, b * n + a])).values
    del a, b
    rp, ci = _csr_from_keys(key, n, device)
    return n, rp, ci, _pattern_values(len(ci), device, dtype)


def wikitalk_like(scale=1.0, device="cpu", dtype=torch.float64):
    """wiki-Talk's shape: 2.39 M x 2.39 M, ~5.0 M non-zeros; 94 % of the rows are empty, the others' lengths follow a power law up to
    100 000 (a handful of rows hold a third of the matrix), columns nearly uniform (everybody gets written to once or twice)"""
    n = max(1024, int(WIKITALK["n"] * scale))
    nnz = max(4096, int(WIKITALK["nnz"] * scale))
    g = _gen(device, WIKITALK["seed"])
    writers = torch.nonzero(torch.rand(n, generator=g, device=device) < 0.06).flatten()
    k = len(writers)
    max_deg = max(16, min(int(100_022 * min(1.0, scale * 4)), n // 2))
    deg = torch.floor((1.0 - torch.rand(k, generator=g, device=device, dtype=torch.float64)) ** (-1.0 / 0.6)).clamp_(1, max_deg)
    deg = (deg * (nnz * 1.02 / float(deg.sum()))).round().clamp_(1, max_deg).to(torch.int64)
    rows = torch.repeat_interleave(writers, deg)
    perm = torch.randperm(n, generator=g, device=device)
    cols = perm[(n * torch.rand(len(rows), generator=g, device=device, dtype=torch.float64) ** 1.3).to(torch.int64).clamp_(0, n - 1)]
    key = torch.unique(rows * n + cols)
    rp, ci = _csr_from_keys(key, n, device)
    return n, rp, ci, _pattern_values(len(ci), device, dtype)

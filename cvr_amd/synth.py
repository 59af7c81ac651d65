"""Seeded synthetic stand-ins for the matrices BASELINE.json names (there is no network: the real
SuiteSparse / SNAP files are used instead when CVR_DATA_DIR holds them).  SURVEY.md 8(d).

All generators return 0-based CSR: (nrows, ncols, row_ptr int64, col_idx int32, vals) with the entries of a
row sorted by column, no duplicates.
"""
import os

import numpy as np

WEB_GOOGLE = dict(n=916_428, nnz=5_105_039, empty_frac=0.193, max_deg=456, seed=20261002)
LIVEJOURNAL = dict(n=4_847_571, nnz=68_993_773, empty_frac=0.11, max_deg=20_293, seed=20261003)


def _coo_to_csr(n, rows, cols, nnz_target=None, rng=None):
    """sort by (row, col), drop duplicates, trim to nnz_target by dropping random entries"""
    key = rows.astype(np.int64) * np.int64(n) + cols.astype(np.int64)
    key = np.unique(key)
    if nnz_target is not None and len(key) > nnz_target:
        drop = rng.choice(len(key), size=len(key) - nnz_target, replace=False)
        keep = np.ones(len(key), dtype=bool)
        keep[drop] = False
        key = key[keep]
    r = (key // n).astype(np.int64)
    c = (key % n).astype(np.int32)
    rp = np.zeros(n + 1, dtype=np.int64)
    rp[1:] = np.cumsum(np.bincount(r, minlength=n))
    return rp, c


def power_law_graph(n, nnz, empty_frac, max_deg, seed, alpha=2.1, local_frac=0.5, local_scale=2000.0, pattern_values=True):
    """Web-graph-like square matrix: Zipf-like out-degrees (rows) with a share of empty rows; half of the
    columns drawn by popularity (power-law in-degree), half near the diagonal (host locality).
    pattern_values: value = (file-order index) % 13, as the reference assigns to `pattern` files
    (spmv.cpp:417); row-major order stands in for the file order."""
    rng = np.random.default_rng(seed)
    nonempty = rng.random(n) >= empty_frac
    k = int(nonempty.sum())
    over = 1.03                                       # duplicates are dropped later
    # discrete power-law degrees >= 1, truncated at max_deg, rescaled to the target mean
    u = rng.random(k)
    deg = np.floor((1.0 - u) ** (-1.0 / (alpha - 1.0))).astype(np.int64)
    deg = np.clip(deg, 1, max_deg)
    want = int(nnz * over)
    scale = want / deg.sum()
    deg = np.clip(np.maximum(1, np.rint(deg * scale)).astype(np.int64), 1, max_deg)
    diff = want - int(deg.sum())
    if diff > 0:                                      # top up on random rows
        idx = rng.integers(0, k, size=diff)
        np.add.at(deg, idx, 1)
        deg = np.clip(deg, 1, max_deg)
    rows_ne = np.nonzero(nonempty)[0]
    rows = np.repeat(rows_ne, deg)
    m = len(rows)
    # columns
    is_local = rng.random(m) < local_frac
    off = np.rint(rng.laplace(0.0, local_scale, size=m)).astype(np.int64)
    loc = np.clip(rows + off, 0, n - 1)
    # popularity: column rank ~ power law, then a fixed permutation scatters the hubs
    pr = rng.random(m)
    rank = np.floor(n * pr ** 2.6).astype(np.int64)   # density ~ rank^(-0.615): heavy head
    perm = rng.permutation(n)
    pop = perm[np.clip(rank, 0, n - 1)]
    cols = np.where(is_local, loc, pop)
    rp, ci = _coo_to_csr(n, rows, cols, nnz_target=nnz, rng=rng)
    if pattern_values:
        vals = (np.arange(len(ci), dtype=np.int64) % 13).astype(np.float64)
    else:
        vals = rng.random(len(ci)) * 2.0 - 1.0
    return n, n, rp, ci, vals


def web_google_like(scale=1.0, seed=None):
    """916 428 x 916 428, 5 105 039 nnz, ~19 % empty rows, max out-degree 456 (the real web-Google's shape);
    `scale` < 1 shrinks rows and nnz together for quick tests."""
    p = dict(WEB_GOOGLE)
    if seed is not None:
        p["seed"] = seed
    n = max(64, int(p["n"] * scale))
    nnz = max(64, int(p["nnz"] * scale))
    return power_law_graph(n, nnz, p["empty_frac"], min(p["max_deg"], n // 2), p["seed"])


def livejournal_like(scale=1.0, seed=None):
    p = dict(LIVEJOURNAL)
    if seed is not None:
        p["seed"] = seed
    n = max(64, int(p["n"] * scale))
    nnz = max(64, int(p["nnz"] * scale))
    return power_law_graph(n, nnz, p["empty_frac"], min(p["max_deg"], n // 2), p["seed"], alpha=1.9, local_scale=50000.0)


def rmat(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=1, dtype=np.float32, dedupe=False):
    """R-MAT (Graph500 parameters): 2^scale vertices, edge_factor * 2^scale edges, duplicates kept (summed
    order = generation order is not kept: entries of a row are sorted by column)."""
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = edge_factor * n
    rows = np.zeros(m, dtype=np.int64)
    cols = np.zeros(m, dtype=np.int64)
    ab, abc = a + b, a + b + c
    for lvl in range(scale):
        r = rng.random(m)
        rbit = r >= ab
        cbit = ((r >= a) & (r < ab)) | (r >= abc)
        rows |= rbit.astype(np.int64) << lvl
        cols |= cbit.astype(np.int64) << lvl
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    if dedupe:
        keep = np.ones(m, dtype=bool)
        keep[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
        rows, cols = rows[keep], cols[keep]
    rp = np.zeros(n + 1, dtype=np.int64)
    rp[1:] = np.cumsum(np.bincount(rows, minlength=n))
    vals = rng.random(len(rows)).astype(dtype)
    return n, n, rp, cols.astype(np.int32), vals


def banded_sym(n, half_band=13, seed=7, dtype=np.float64):
    """KKT-like banded symmetric pattern (~2*half_band+1 nnz per row): the stand-in for nlpkkt240's shape"""
    rng = np.random.default_rng(seed)
    offs = np.arange(-half_band, half_band + 1, dtype=np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), len(offs))
    cols = rows + np.tile(offs, n)
    ok = (cols >= 0) & (cols < n)
    rows, cols = rows[ok], cols[ok]
    rp = np.zeros(n + 1, dtype=np.int64)
    rp[1:] = np.cumsum(np.bincount(rows, minlength=n))
    vals = (rng.random(len(rows)) * 2 - 1).astype(dtype)
    return n, n, rp, cols.astype(np.int32), vals


def x_rand(n, dtype=np.float64):
    """splitmix64(0xC0FFEE, j) -> uniform [-1, 1): identical to cvr_fill_x(mode 1) and the oracle's orc_x_rand"""
    j = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(0xC0FFEE) + (j + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0).astype(dtype)


def b_alg(nrows, ncols, nnz, vbytes=8):
    """algorithmic bytes of one SpMV, each array touched once"""
    return nnz * (vbytes + 4) + (nrows + 1) * 4 + ncols * vbytes + nrows * vbytes


def to_refcompat(nrows, ncols, rp, ci, vals):
    """0-based CSR -> the reference loader's 1-based int arrays as it would have
    produced them from a Matrix-Market file of these entries: nnz padded to a multiple of 16 with zero copies
    of the last entry, rowptr[numRows+2] with tail = nItems-1.  For bench.py's CPU baseline."""
    nnz = len(ci)
    npad = nnz if nnz % 16 == 0 else (nnz + 16) // 16 * 16
    rows = np.repeat(np.arange(nrows, dtype=np.int64), np.diff(rp))
    last_r, last_c = int(rows[-1]), int(ci[-1])
    cols1 = np.concatenate([ci.astype(np.int32) + 1, np.full(npad - nnz, last_c + 1, dtype=np.int32)])
    v = np.concatenate([np.asarray(vals, dtype=np.float32).astype(np.float64), np.zeros(npad - nnz)])
    cnt = np.bincount(rows, minlength=nrows).astype(np.int64)
    cnt[last_r] += npad - nnz
    # the pads (same coordinates as the last entry) sort to the end of the last non-empty row
    rp1 = np.zeros(nrows + 2, dtype=np.int64)      # 1-based rows: row 0 is empty (Q1)
    rp1[2:] = np.cumsum(cnt)
    rp1[last_r + 2:] = npad - 1                    # Q9: row pointers after the last non-empty row
    return dict(nItems=npad, nItemsRaw=nnz, numRows=nrows, numCols=ncols, val=v, cols=cols1, rowptr=rp1.astype(np.int32))


def data_file(name):
    d = os.environ.get("CVR_DATA_DIR")
    if d and os.path.exists(os.path.join(d, name)):
        return os.path.join(d, name)
    return None

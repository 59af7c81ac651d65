"""Seeded small matrices for the parity tests: the adversarial shapes of SURVEY.md section 4 (rows spanning
several chunks, chunks with fewer rows than lanes, empty rows everywhere, single row, dense row + singletons)."""
import numpy as np


def csr_from_lengths(lens, ncols, rng, dtype=np.float64, sort=True):
    lens = np.asarray(lens, dtype=np.int64)
    rp = np.zeros(len(lens) + 1, dtype=np.int64)
    rp[1:] = np.cumsum(lens)
    ci = rng.integers(0, ncols, size=int(rp[-1])).astype(np.int32)
    if sort:
        for r in range(len(lens)):
            ci[rp[r]:rp[r + 1]].sort()
    va = (rng.random(int(rp[-1])) * 2 - 1).astype(dtype)
    return len(lens), ncols, rp, ci, va


def cases(dtype=np.float64):
    rng = np.random.default_rng(20261002)
    out = {}
    out["empty_matrix_rows_only"] = csr_from_lengths([0] * 37, 5, rng, dtype)
    out["single_entry"] = csr_from_lengths([1], 1, rng, dtype)
    out["one_row_long"] = csr_from_lengths([5000], 300, rng, dtype)                      # one row over many chunks
    out["diag_96"] = csr_from_lengths([1] * 96, 96, rng, dtype)
    out["exact_fill"] = csr_from_lengths([8] * 64, 64, rng, dtype)                       # 512 slots = one S=8 chunk
    out["few_rows_lt_lanes"] = csr_from_lengths([300, 2, 1, 700, 3], 1000, rng, dtype)    # fewer rows than lanes
    out["leading_trailing_empty"] = csr_from_lengths([0] * 70 + [3, 0, 0, 9, 1] * 40 + [0] * 130, 500, rng, dtype)
    out["dense_row_plus_singletons"] = csr_from_lengths([1] * 500 + [20000] + [1] * 500, 4096, rng, dtype)
    out["two_giants"] = csr_from_lengths([3000, 0, 0, 4097, 5] + [2] * 50, 2048, rng, dtype)
    lens = np.minimum((rng.pareto(1.3, size=3000) + 1).astype(np.int64), 900)
    lens[rng.random(3000) < 0.25] = 0
    out["power_law_3000"] = csr_from_lengths(lens, 3000, rng, dtype)
    lens = rng.integers(0, 40, size=2000)
    out["uniform_2000"] = csr_from_lengths(lens, 777, rng, dtype)
    out["thr_edge"] = csr_from_lengths([127, 128, 129, 130, 1, 126, 2, 128, 128, 128, 128, 1], 64, rng, dtype)
    return out

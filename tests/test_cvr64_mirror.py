"""CPU: the CVR64 format (oracle/cvr64_mirror.c, written with the reference's sequential refill semantics,
spmv.cpp:814-946) against the CSR oracle (spmv.cpp:1843-1850), and the product's host planner
(cvr_amd/csrc/cvr_plan.cpp, through the C ABI) against the mirror's plan.  No GPU."""
import os

import numpy as np
import pytest

import cases as K
import oraclelib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = K.cases()
CASES32 = K.cases(np.float32)


def _check(nrows, ncols, rp, ci, va, S, thr=0):
    m = O.Cvr64(nrows, ncols, rp, ci, va, S, thr)
    f32 = va.dtype == np.float32
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode).astype(va.dtype)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y = m.spmv(x)
        bad, worst = O.tol_check(y, yref, absy, tol=1e-5 if f32 else 1e-12)
        if f32:
            bad = bad[np.abs(np.asarray(y, dtype=np.float64) - yref)[bad] > 1e-6 * np.maximum(1.0, absy[bad])]
        assert len(bad) == 0, (mode, S, worst, bad[:5])
    return m


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S", [4, 8, 32])
def test_mirror_matches_csr_oracle(name, S):
    m = _check(*CASES[name], S)
    nrows = CASES[name][0]
    # structure: every chunk holds exactly 64*S slots; descriptors stay inside y_ext
    assert m.image.size == m.nchunks * (S // 4) * 3072
    assert np.all(m.desc[:, 2] < nrows + 1 + 2 * m.nchunks) and np.all(m.desc[:, 3] < nrows + 1 + 2 * m.nchunks)


@pytest.mark.parametrize("name", ["few_rows_lt_lanes", "power_law_3000", "two_giants", "leading_trailing_empty"])
def test_mirror_fp32(name):
    _check(*CASES32[name], 8)


@pytest.mark.parametrize("thr", [1, 16, 100000])
def test_mirror_split_threshold(thr):
    _check(*CASES["power_law_3000"], 8, thr)
    _check(*CASES["two_giants"], 16, thr)


def test_mirror_is_a_permutation_of_csr():
    """CVR is a permutation of the CSR non-zeros plus explicit pads (property 1)"""
    nrows, ncols, rp, ci, va = CASES["power_law_3000"]
    m = O.Cvr64(nrows, ncols, rp, ci, va, 8)
    img = m.image.reshape(m.nchunks * 2, 3072)
    cols = img[:, :1024].copy().view(np.uint32).reshape(-1, 64, 4)
    vals = img[:, 1024:].copy().view(np.float64).reshape(-1, 2, 64, 2)
    c = (cols & 0x7FFFFFFF).reshape(-1)
    v = np.stack([vals[:, 0, :, 0], vals[:, 0, :, 1], vals[:, 1, :, 0], vals[:, 1, :, 1]], axis=-1).reshape(-1)
    real = c != ncols
    assert real.sum() == len(ci)
    assert np.array_equal(np.sort(c[real].astype(np.int64) * 4 + 0), np.sort(ci.astype(np.int64) * 4 + 0))
    assert np.array_equal(np.sort(v[real]), np.sort(va))
    assert np.all(v[~real] == 0)
    # one end flag per segment (rows with a segment here + pad segments) and one per stolen piece
    stolen = (m.target != np.arange(64, dtype=np.uint8)[None, :]).sum()
    assert (cols >> 31).sum() == m.desc[:, 1].sum() + stolen


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S", [4, 16])
def test_product_planner_equals_mirror(name, S):
    import cvr_amd
    nrows, ncols, rp, ci, va = CASES[name]
    m = O.Cvr64(nrows, ncols, rp, ci, va, S)
    p = cvr_amd.plan_chunks(rp, S)
    assert len(p["row_first"]) == m.nchunks
    assert np.array_equal(p["nz_begin"], m.nz_begin)
    assert np.array_equal(p["row_first"], m.desc[:, 0].astype(np.int64))
    assert np.array_equal(p["nseg"], m.desc[:, 1].astype(np.int64))
    assert np.array_equal(p["pad_cnt"], m.pad_cnt)


def test_mirror_on_reference_loader_arrays():
    """the literal 1-based arrays of the reference loader (Q1, Q9) through CVR64 == the reference's CSR y"""
    for name in ("pl2000_pattern", "skew12", "sym250_real", "rect64x300_int", "onerow"):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        nItems, numRows, numCols = (int(v) for v in z["dims"])
        rp = z["csr_rowptr"].astype(np.int64)
        nrows, ncols = numRows + 1, numCols + 1
        m = O.Cvr64(nrows, ncols, rp, z["csr_col"], z["csr_val"], 8)
        for mode in ("ones", "rand"):
            x = z[f"x_{mode}"][:ncols]
            y = m.spmv(x)
            yref = z[f"y_csr_{mode}"]
            _, absy = O.csr_spmv64(rp, z["csr_col"], z["csr_val"], x)
            bad, worst = O.tol_check(y[:numRows], yref, absy[:numRows])
            assert len(bad) == 0, (name, mode, worst)


@pytest.mark.parametrize("name", ["power_law_3000", "two_giants", "leading_trailing_empty", "single_entry"])
def test_mirror_value_dictionary(name):
    """the dictionary layout (one code byte per slot) of the mirror against the CSR oracle; > 256 values are refused"""
    nrows, ncols, rp, ci, va = CASES[name]
    vq = ((np.arange(len(va)) % 13) - 3).astype(np.float64)
    for S in (4, 16):
        m = O.Cvr64(nrows, ncols, rp, ci, vq, S, use_dict=True)
        assert m.image.size == m.nchunks * (S // 4) * 1280 and 1 <= m.ndict <= 14
        x = O.x_vec_fast(ncols, "rand")
        yref, absy = O.csr_spmv64(rp, ci, vq, x)
        bad, worst = O.tol_check(m.spmv(x), yref, absy)
        assert len(bad) == 0, (name, S, worst)
    if len(np.unique(va)) > 256:
        with pytest.raises(RuntimeError):
            O.Cvr64(nrows, ncols, rp, ci, va, 8, use_dict=True)


def test_product_planner_equals_mirror_on_random_row_lengths():
    """the product's host planner (eight-rows-at-a-time fast path, split threshold, pad segments) against the mirror's
    one-row-at-a-time planner on random row-length distributions: same chunk starts, first rows, segment and pad counts;
    and the mirror built on that plan still reproduces the CSR y"""
    import cvr_amd
    rng = np.random.default_rng(424242)
    for case in range(120):
        kind = case % 6
        nrows = int(rng.integers(1, 2500))
        if kind == 0:
            lens = rng.integers(0, 3, nrows)
        elif kind == 1:
            lens = np.minimum((rng.pareto(1.0, nrows) + 0.5).astype(np.int64), 4000)
        elif kind == 2:
            lens = np.full(nrows, int(rng.integers(1, 9)))                     # exact fills of 8-row blocks
        elif kind == 3:
            lens = np.where(rng.random(nrows) < 0.02, rng.integers(500, 9000, nrows), rng.integers(0, 4, nrows))
        elif kind == 4:
            lens = np.zeros(nrows, dtype=np.int64)                             # only empty rows (pad slots)
        else:
            lens = rng.integers(0, 40, nrows) * (rng.random(nrows) < 0.5)
        lens = np.asarray(lens, dtype=np.int64)
        ncols = int(rng.integers(1, 3000))
        nrows, ncols, rp, ci, va = K.csr_from_lengths(lens, ncols, rng)
        S = int(rng.choice([4, 8, 12, 16, 28, 32, 56, 64, 128]))
        thr = int(rng.choice([0, 1, 5, 64, 10**7]))
        m = O.Cvr64(nrows, ncols, rp, ci, va, S, thr)
        p = cvr_amd.plan_chunks(rp, S, thr)
        ctx = dict(case=case, kind=kind, nrows=nrows, S=S, thr=thr)
        assert len(p["row_first"]) == m.nchunks, ctx
        assert np.array_equal(p["nz_begin"], m.nz_begin), ctx
        assert np.array_equal(p["row_first"], m.desc[:, 0].astype(np.int64)), ctx
        assert np.array_equal(p["nseg"], m.desc[:, 1].astype(np.int64)), ctx
        assert np.array_equal(p["pad_cnt"], m.pad_cnt), ctx
        x = O.x_vec_fast(ncols, "rand")
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        bad, worst = O.tol_check(m.spmv(x), yref, absy + 1e-30)
        assert len(bad) == 0, (ctx, worst)


@pytest.mark.parametrize("S,thr", [(8, 0), (32, 3)])
def test_row_block_restarts_product_planner_equals_mirror(S, thr):
    """more than 65 536 rows: the planner restarts at every row block (the blocks are planned by parallel threads in the
    product, one after the other in the mirror): same chunks, and the mirror's y still equals the CSR oracle's"""
    import cvr_amd
    rng = np.random.default_rng(77)
    lens = rng.integers(0, 9, size=150_000)
    lens[rng.integers(0, len(lens), size=40)] = rng.integers(300, 3000, size=40)       # rows cut over chunks, some near block ends
    lens[65535] = 2500
    lens[131071] = 0
    nrows, ncols, rp, ci, va = K.csr_from_lengths(lens, 5000, rng, sort=False)
    m = O.Cvr64(nrows, ncols, rp, ci, va, S, thr)
    p = cvr_amd.plan_chunks(rp, S, thr)
    assert len(p["row_first"]) == m.nchunks
    assert np.array_equal(p["nz_begin"], m.nz_begin) and np.array_equal(p["pad_cnt"], m.pad_cnt)
    assert np.array_equal(p["row_first"], m.desc[:, 0].astype(np.int64)) and np.array_equal(p["nseg"], m.desc[:, 1].astype(np.int64))
    starts = set(p["row_first"].tolist())
    assert 65536 in starts and 131072 in starts                       # a chunk begins at every block boundary
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    bad, worst = O.tol_check(m.spmv(x), yref, absy)
    assert len(bad) == 0, worst


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S,P,tags,pmax", [(8, 3, 0, 0), (32, 5, 0, 4), (8, 3, 1, 1), (64, 7, 1, 8)])
def test_mirror_column_phases_match_csr_oracle(name, S, P, tags, pmax):
    """column phases: every piece of a lane stream (a (row, phase) segment, or what a lane stole of one) carries its row -- above
    the column index, or in a 16-bit tag of its own -- and adds its sum to that row; the mirror's y against the CSR oracle"""
    nrows, ncols, rp, ci, va = CASES[name]
    if ncols < 64 * P:
        pytest.skip("too few columns for that many phases")
    m = O.Cvr64(nrows, ncols, rp, ci, va, S, phases=P, max_rows=min(64 * S, 2000), tag16=tags, piece_max=pmax)
    gb = 3072 + (512 if tags else 0)
    assert m.image.size == m.nchunks * (S // 4) * gb
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        bad, worst = O.tol_check(m.spmv(x), yref, absy, tol=1e-12)
        assert len(bad) == 0, (mode, worst, bad[:5])
    if tags:      # the two encodings hold the same pieces: the column words differ only in the row field
        m0 = O.Cvr64(nrows, ncols, rp, ci, va, S, phases=P, max_rows=min(64 * S, 2000), tag16=0, piece_max=pmax)
        assert np.array_equal(m.desc, m0.desc) and np.array_equal(m.target, m0.target)
        x = O.x_vec_fast(ncols, "rand")
        assert np.array_equal(m.spmv(x), m0.spmv(x))


# ---- interleaved chunks (cvr_options.interleave; oracle: orc_cvr64_build_ilv) ----
@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S,tags,use_dict", [(4, 1, 0), (16, 1, 1), (32, 0, 0)])
def test_interleaved_mirror_matches_csr_oracle(name, S, tags, use_dict):
    """the interleaved image (a chunk's non-zeros dealt to the lanes in column order, every slot with its row) gives the CSR loop's y,
    and is a permutation of the CSR non-zeros plus pad slots that point at the dump entry"""
    nrows, ncols, rp, ci, va = CASES[name]
    if use_dict and len(np.unique(va)) > 255:
        pytest.skip("more than 255 distinct values")
    max_rows = min(64 * S, 2000)
    if not tags:
        bits = max(1, int(ncols).bit_length())
        max_rows = min(max_rows, (1 << (32 - bits)) - 1)          # (no end flag in an interleaved column word: bits [col_bits, 32) hold the row)
    m = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=bool(use_dict), max_rows=max_rows, tag16=tags, interleave=True)
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        bad, worst = O.tol_check(m.spmv(x), yref, absy, tol=1e-12)
        assert len(bad) == 0, (mode, worst, bad[:5])
    gb = (1280 if use_dict else 3072) + (512 if tags else 0)
    img = m.image.reshape(-1, gb)
    cw = img[:, :1024].copy().view(np.uint32).reshape(m.nchunks, -1, 64, 4)          # [chunk][group][lane][step in group]
    if tags:
        assert np.all(cw >> 31 == 1)                                                  # (with tags the column word keeps the end flag: every slot ends a piece)
    cmask = 0x7FFFFFFF if tags else (1 << m.c.col_bits) - 1
    col = (cw & cmask).transpose(0, 1, 3, 2).reshape(m.nchunks, -1)                   # [chunk][slot in (step, lane) order]
    real = col != ncols
    assert real.sum() == len(ci)
    for k in range(m.nchunks):
        n = int(m.nz_begin[k + 1] - m.nz_begin[k])
        assert np.all(real[k, :n]) and not real[k, n:].any()                          # the chunk's non-zeros first, then padding
        assert np.all(np.diff(col[k, :n].astype(np.int64)) >= 0)                      # in column order
        assert np.array_equal(np.sort(col[k, :n]), np.sort(ci[m.nz_begin[k]:m.nz_begin[k + 1]]).astype(np.uint32))


def test_interleaved_mirror_fp32_and_limits():
    nrows, ncols, rp, ci, va = CASES32["power_law_3000"]
    m = O.Cvr64(nrows, ncols, rp, ci, va, 16, max_rows=500, tag16=1, interleave=True)
    x = O.x_vec_fast(ncols, "rand").astype(np.float32)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    bad, worst = O.tol_check(m.spmv(x), yref, absy, tol=1e-5)
    assert len(bad) == 0, worst
    with pytest.raises(RuntimeError):          # the row field of the column word cannot hold that many rows
        O.Cvr64(nrows, ncols, rp, ci, va, 16, max_rows=1 << 20, tag16=0, interleave=True)


# ---- gang chunks (cvr_options.gang; oracle: orc_cvr64_build_gang) ----
@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S,gang,tags,use_dict", [(4, 4, 0, 0), (16, 4, 1, 1), (32, 2, 0, 1), (16, 8, 0, 0)])
def test_gang_mirror_matches_csr_oracle(name, S, gang, tags, use_dict):
    """gang chunks: the chunks of a workgroup sorted together -- the mirror's image gives the CSR loop's y, every gang's list is in column order and a
    permutation of its chunks' non-zeros, the groups' first columns are what the offsets are relative to, the tags name (chunk, row) of the plan"""
    nrows, ncols, rp, ci, va = CASES[name]
    if use_dict and len(np.unique(va)) > 255:
        pytest.skip("more than 255 distinct values")
    ystage = 512
    max_rows = min(64 * S, ystage - 1)
    try:
        m = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=bool(use_dict), max_rows=max_rows, tag16=tags, gang=gang, ystage=ystage)
    except RuntimeError as e:
        assert "-9" in str(e) and not tags          # a column further than 2^17 from its group's first: the product falls back to tags
        m = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=bool(use_dict), max_rows=max_rows, tag16=1, gang=gang, ystage=ystage)
        tags = 1
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        bad, worst = O.tol_check(m.spmv(x), yref, absy, tol=1e-12)
        assert len(bad) == 0, (mode, worst, bad[:5])
    G = S // 4
    gb = (1280 if use_dict else 3072) + (512 if tags else 0)
    img = m.image.reshape(-1, gb)
    cw = img[:, :1024].copy().view(np.uint32).reshape(-1, 64, 4)                       # [group of the image][lane][step]
    seen = 0
    for k0 in range(0, m.nchunks, gang):
        k1 = min(k0 + gang, m.nchunks)
        n = int(m.nz_begin[k1] - m.nz_begin[k0])
        gg = int(m.ggroups[k0])
        assert gg == (n + 255) // 256 and gg <= (k1 - k0) * G
        w = cw[k0 * G:k0 * G + gg].transpose(0, 2, 1).reshape(-1)                      # the gang's slots in list order
        if tags:
            col = (w & 0x7FFFFFFF).astype(np.int64)
            tg = img[k0 * G:k0 * G + gg, 1024:1536].copy().view(np.uint16).reshape(-1, 64, 4).transpose(0, 2, 1).reshape(-1).astype(np.int64)
        else:
            col = (w & 0x1FFFF).astype(np.int64) + np.repeat(m.gbase[k0 * G:k0 * G + gg].astype(np.int64), 256)
            tg = (w >> 17).astype(np.int64)
        assert np.all(np.diff(col[:n]) >= 0)                                           # column order
        assert np.array_equal(np.sort(col[:n]), np.sort(ci[m.nz_begin[k0]:m.nz_begin[k1]]).astype(np.int64))
        assert np.all(tg[:n] // ystage < k1 - k0)
        rows_of = m.desc[k0:k1, 0].astype(np.int64)[tg[:n] // ystage] + tg[:n] % ystage   # the rows the tags name
        want = np.repeat(np.arange(nrows), np.diff(rp))[m.nz_begin[k0]:m.nz_begin[k1]]
        assert np.array_equal(np.sort(rows_of), np.sort(want))
        seen += n
    assert seen == len(ci)


def test_gang_mirror_fp32_and_limits():
    nrows, ncols, rp, ci, va = CASES32["power_law_3000"]
    m = O.Cvr64(nrows, ncols, rp, ci, va, 16, max_rows=500, tag16=0, gang=4, ystage=512)
    x = O.x_vec_fast(ncols, "rand").astype(np.float32)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    bad, worst = O.tol_check(m.spmv(x), yref, absy, tol=1e-5)
    assert len(bad) == 0, worst
    with pytest.raises(RuntimeError):          # the tags of a gang have 15 bits: 8 chunks x 8 192 accumulators do not fit
        O.Cvr64(nrows, ncols, rp, ci, va, 16, max_rows=500, tag16=0, gang=8, ystage=8192)
    with pytest.raises(RuntimeError):          # a chunk's rows must leave room for its dump entry
        O.Cvr64(nrows, ncols, rp, ci, va, 16, max_rows=512, tag16=0, gang=4, ystage=512)

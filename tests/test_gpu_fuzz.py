"""GPU: randomised parity -- random row-length distributions (empty rows, giants, uniform, power law), random chunk
length, split threshold, column panels, waves per workgroup, LDS window and column phases; converter image against the CPU mirror (bit for bit, when one
image) and y against the CSR oracle.  CVR_FUZZ_CASES sets the number of cases (default 160, and 96 more through the device planner)."""
import os

import numpy as np
import pytest

import cases as K
import cvr_amd
import oraclelib as O

pytestmark = pytest.mark.gpu

try:
    import torch
except Exception:      # noqa: BLE001 -- the device-array cases are skipped without torch
    torch = None


def _random_case(rng):
    kind = rng.integers(0, 5)
    nrows = int(rng.integers(1, 4000))
    ncols = int(rng.integers(1, 6000))
    if kind == 0:
        lens = rng.integers(0, 12, size=nrows)
    elif kind == 1:
        lens = np.minimum((rng.pareto(1.1, size=nrows) + 0.5).astype(np.int64), 3000)
    elif kind == 2:
        lens = np.zeros(nrows, dtype=np.int64)
        lens[rng.integers(0, nrows, size=max(1, nrows // 50))] = rng.integers(1, 5000, size=max(1, nrows // 50))
    elif kind == 3:
        lens = np.full(nrows, int(rng.integers(1, 70)))
    else:
        lens = rng.integers(0, 3, size=nrows) * rng.integers(0, 40, size=nrows)
    lens = np.asarray(lens, dtype=np.int64)
    if lens.sum() > 400_000:
        lens = lens // (lens.sum() // 400_000 + 1)
    srt = bool(rng.integers(0, 2))
    return K.csr_from_lengths(lens, ncols, rng, sort=srt) + (srt,)


def test_fuzz_parity_through_the_device_planner():
    """the same cases with every matrix planned on the device (cvr_plan_dev.hip; cvr_create uses it from 200 000 rows on):
    the image is still the mirror's bit for bit"""
    os.environ["CVR_DEBUG"] = "device_plan_rows=0"
    try:
        test_fuzz_parity(int(os.environ.get("CVR_FUZZ_CASES", "96")), 20261003)
    finally:
        del os.environ["CVR_DEBUG"]


def test_fuzz_parity(ncases=None, seed=None):
    ncases = int(os.environ.get("CVR_FUZZ_CASES", "160")) if ncases is None else ncases
    rng = np.random.default_rng(int(os.environ.get("CVR_FUZZ_SEED", "20261002")) if seed is None else seed)
    for case in range(ncases):
        nrows, ncols, rp, ci, va, srt = _random_case(rng)
        f32 = bool(rng.integers(0, 4) == 0)
        if rng.integers(0, 3) == 0:                      # few distinct values: the value dictionary kicks in
            va = rng.choice(np.array([0.0, 1.0, -1.0, 0.5, 3.0, 1e-3, -7.25]), size=len(va))
        if f32:
            va = va.astype(np.float32)
        S = int(rng.choice([4, 8, 12, 16, 32, 64, 96, 128]))
        thr = int(rng.choice([0, 1, 7, 64, 10**6]))
        P = int(rng.choice([1, 1, 2, 3]))
        win = int(rng.choice([0, 0, 64, 1000]))
        wpb = int(rng.choice([1, 1, 2, 8, 16]))
        ph = int(rng.choice([1, 1, 2, 5])) if P == 1 and ncols >= 320 else 1
        hub = int(rng.choice([0, 0, 7, 300])) if ph == 1 else 0
        reo = int(rng.integers(0, 2)) if hub else 0
        tags = int(rng.choice([-1, 0, 1])) if ph > 1 else -1        # wide row tags, bounded pieces (column phases only)
        pmax = int(rng.choice([-1, 0, 2, 8])) if ph > 1 else -1
        ctx = dict(case=case, nrows=nrows, ncols=ncols, nnz=len(ci), S=S, thr=thr, P=P, win=win, f32=f32, wpb=wpb, phases=ph, sorted=srt, hub=hub, tags=tags, pmax=pmax)
        from_dev = torch is not None and len(ci) > 0 and rng.integers(0, 4) == 0     # CSR arrays already on the device
        if os.environ.get("CVR_FUZZ_TRACE"):
            print(ctx, "from_dev", from_dev, flush=True)
        if from_dev:
            keep = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rp.astype(np.int64), ci.astype(np.int32), va)]
            torch.cuda.synchronize()
            A = cvr_amd.CvrMatrix.from_device(nrows, ncols, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), is_f32=f32,
                                              steps_per_chunk=S, split_threshold=thr, col_panels=P)
            win, ph, hub = 0, 1, 0
        else:
            if ph > 1 and not srt and len(ci) > 0 and np.any((np.diff(ci.astype(np.int64)) < 0) & (np.diff(np.repeat(np.arange(nrows), np.diff(rp))) == 0)):
                with pytest.raises(cvr_amd.CvrError):      # column phases need ascending columns inside every row
                    cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, split_threshold=thr, col_panels=P, x_window=win,
                                      waves_per_block=wpb, col_phases=ph)
                ph = 1
            if ph > 1:
                hub = 0
            A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, split_threshold=thr, col_panels=P, x_window=win,
                                  waves_per_block=wpb, col_phases=ph, hub_table=hub, hub_reorder=reo if P == 1 else 0, row_tags16=tags, piece_max=pmax,
                                  xcd_swizzle=int(rng.choice([-1, -1, 0, 3, 4, 5, 6])))
            ph = A.info.col_phases
        ctx["from_dev"] = bool(from_dev)
        if P == 1:
            mir = O.Cvr64(nrows, ncols, rp, ci, va, S, thr, use_dict=A.info.value_dict > 0, phases=ph, max_rows=A.info.chunk_row_cap, hub_max=A.info.hub_entries, narrow=A.info.narrow_cols, reorder=A.info.hub_reorder,
                          tag16=A.info.row_tags16, piece_max=A.info.piece_max)
            img = A.export_image()
            assert np.array_equal(img["image"], mir.image) and np.array_equal(img["desc"], mir.desc), ctx
            assert np.array_equal(img["target"], mir.target) and np.array_equal(img["shared"], mir.shared), ctx
        x = O.x_vec_fast(ncols, "rand").astype(va.dtype)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        bad, worst = O.tol_check(y, yref, absy + 1e-30, tol=1e-5 if f32 else 1e-12)
        assert len(bad) == 0, (ctx, bad[:5], worst)
        y2, _ = A.spmv(x)
        assert np.array_equal(y.view(np.uint8), y2.view(np.uint8)), ctx
        A.close()


def test_fuzz_interleaved_chunks():
    """cvr_options.interleave over the same random shapes (sorted or unsorted rows, host or device planner): one image -> the CPU mirror's
    bits (orc_cvr64_build_ilv) and y of the mirror; column panels -> y against the CSR oracle; reruns bit for bit"""
    ncases = int(os.environ.get("CVR_FUZZ_CASES", "120"))
    rng = np.random.default_rng(int(os.environ.get("CVR_FUZZ_SEED", "20261006")))
    for case in range(ncases):
        nrows, ncols, rp, ci, va, srt = _random_case(rng)
        f32 = bool(rng.integers(0, 4) == 0)
        if rng.integers(0, 2) == 0:
            va = rng.choice(np.array([0.0, 1.0, -1.0, 0.5, 3.0, 1e-3, -7.25]), size=len(va))
        va = va.astype(np.float32 if f32 else np.float64)
        S = int(rng.choice([4, 8, 16, 32, 64, 128, 256, 508]))
        P = int(rng.choice([1, 1, 1, 2, 3, 8]))
        wpb = int(rng.choice([0, 1, 2, 4, 8]))
        tags = int(rng.choice([-1, -1, 0, 1]))
        dev_plan = bool(rng.integers(0, 3) == 0)
        gang = int(rng.integers(0, 2))          # gang chunks: the workgroup's chunks sorted together (takes effect with two wavefronts or more)
        ctx = dict(case=case, nrows=nrows, ncols=ncols, nnz=len(ci), S=S, P=P, wpb=wpb, tags=tags, f32=f32, srt=srt, dev_plan=dev_plan, gang=gang)
        if dev_plan:
            os.environ["CVR_DEBUG"] = "device_plan_rows=0"
        try:
            try:
                A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, col_panels=P, waves_per_block=wpb, row_tags16=tags, interleave=1, gang=gang)
            except cvr_amd.CvrError as e:          # (tags forced off where column and row do not fit one word: a refusal with a code, not a wrong image)
                assert tags == 0, (ctx, str(e))
                continue
        finally:
            os.environ.pop("CVR_DEBUG", None)
        i = A.info
        assert i.interleave == 1, ctx
        x = O.x_vec_fast(ncols, "rand").astype(va.dtype)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        bad, worst = O.tol_check(y, yref, absy + 1e-30, tol=1e-5 if f32 else 1e-12)
        assert len(bad) == 0, (ctx, bad[:5], worst)
        assert (i.gang > 0) == (gang > 0 and i.waves_per_block >= 2), ctx
        if i.col_panels == 1:
            if i.gang:
                vs = 4 if f32 else 8
                ystage = (i.lds_bytes - 80 - (256 * vs if i.value_dict else 0)) // (i.waves_per_block * vs)
                mir = O.Cvr64(nrows, ncols, rp, ci, va, i.steps_per_chunk, use_dict=i.value_dict > 0, max_rows=i.chunk_row_cap, tag16=i.row_tags16, gang=i.gang, ystage=ystage)
            else:
                mir = O.Cvr64(nrows, ncols, rp, ci, va, i.steps_per_chunk, use_dict=i.value_dict > 0, max_rows=i.chunk_row_cap, tag16=i.row_tags16, interleave=True)
            img = A.export_image()
            assert np.array_equal(img["image"], mir.image) and np.array_equal(img["desc"], mir.desc) and np.array_equal(img["shared"], mir.shared), ctx
            if i.gang:
                assert np.array_equal(img["gbase"], mir.gbase) and np.array_equal(img["desc2"][:, 0], mir.ggroups), ctx
            if i.nshared == 0:
                assert np.array_equal(y.view(np.uint8), mir.spmv(x).view(np.uint8)), ctx
        y2, _ = A.spmv(x)
        assert np.array_equal(y.view(np.uint8), y2.view(np.uint8)), ctx
        A.close()


def test_fuzz_one_submission_preprocessing():
    """Matrices of the size the automatic layout makes resident (350 000 - 600 000 rows; local and far columns mixed, empty rows, a few
    rows far beyond a chunk, square and rectangular, fp64 / fp32, few or many distinct values): cvr_create's one-submission path
    against the staged one -- same layout, same image and descriptors bit for bit, same y bits -- and y against the CSR oracle."""
    import scipy.sparse as sp
    rng = np.random.default_rng(int(os.environ.get("CVR_FUZZ_SEED", "20261004")))
    taken = 0
    for case in range(int(os.environ.get("CVR_FUZZ_BIG_CASES", "5"))):
        f32 = bool(rng.integers(0, 3) == 0)
        nrows = int(rng.integers(700_000, 900_000)) if f32 else int(rng.integers(350_000, 600_000))      # (x beyond 2.5 MB: column phases)
        ncols = nrows if rng.integers(0, 3) else nrows + int(rng.integers(1, 200_000))
        deg = rng.zipf(2.2, nrows).clip(0, 400)
        deg[rng.random(nrows) < rng.choice([0.0, 0.05, 0.3])] = 0                 # empty rows
        for r in rng.integers(0, nrows, int(rng.integers(0, 4))):
            deg[r] = int(rng.integers(3_000, 12_000))                             # rows cut over several chunks
        deg = (deg * (2_600_000 / max(deg.sum(), 1))).astype(np.int64).clip(0, ncols // 2) if deg.sum() < 2_600_000 else deg
        rows = np.repeat(np.arange(nrows), deg)
        local = rng.random(len(rows)) < rng.choice([0.3, 0.5, 0.7])
        cols = np.where(local, np.clip(rows + rng.normal(0, 1500, len(rows)).astype(np.int64), 0, ncols - 1), rng.integers(0, ncols, len(rows)))
        few = bool(rng.integers(0, 2))
        vals = rng.choice(np.array([1.0, 2.0, -1.0, 0.5, 3.0, 1e-3, -7.25]), size=len(rows)) if few else rng.standard_normal(len(rows))
        M = sp.csr_matrix((vals, (rows, cols)), shape=(nrows, ncols))
        M.sum_duplicates(); M.sort_indices()
        rp, ci = M.indptr.astype(np.int64), M.indices.astype(np.int32)
        va = M.data.astype(np.float32 if f32 else np.float64)
        ctx = dict(case=case, nrows=nrows, ncols=ncols, nnz=len(ci), f32=f32, few=few)
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
        os.environ["CVR_DEBUG"] = "no_fused"
        try:
            B = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
        finally:
            del os.environ["CVR_DEBUG"]
        ia, ib = A.info, B.info
        taken += ia.preprocess_fused
        for f in ("steps_per_chunk", "waves_per_block", "x_window", "col_phases", "value_dict", "nchunks", "nshared", "nsegments", "row_tags16", "piece_max", "chunk_row_cap"):
            assert getattr(ia, f) == getattr(ib, f), (f, ctx)
        ea, eb = A.export_image(), B.export_image()
        for key in ("desc", "target", "shared", "image"):
            assert np.array_equal(ea[key], eb[key]), (key, ctx)
        x = O.x_vec_fast(ncols, "rand").astype(va.dtype)
        ya, _ = A.spmv(x)
        yb, _ = B.spmv(x)
        assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8)), ctx
        yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
        tol = 1e-5 if f32 else 1e-12
        assert np.all(np.abs(ya.astype(np.float64) - yref) <= tol * absy + 1e-300), ctx
        A.close(); B.close()
    assert taken >= 3                      # (most of the cases are of the resident kind: otherwise the test tests nothing)

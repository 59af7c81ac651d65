"""GPU: the device planner (cvr_amd/csrc/cvr_plan_dev.hip) against the host planner (cvr_plan.cpp, itself pinned against the
mirror's one-row-at-a-time planner in test_cvr64_mirror.py): same chunks and same cut rows, field by field, on random
row-length distributions, with and without a row cap, across row-block restarts."""
import numpy as np
import pytest

try:                   # before the first call into libcvr_amd: PyTorch brings its own copy of the HIP runtime, and whichever copy
    import torch       # is initialised second in a process finds no GPU
except Exception:      # noqa: BLE001 -- the device-array case is skipped without torch
    torch = None

import cvr_amd

pytestmark = pytest.mark.gpu


def _lens(rng, kind, nrows):
    if kind == 0:
        return rng.integers(0, 3, nrows)
    if kind == 1:
        return np.minimum((rng.pareto(1.0, nrows) + 0.5).astype(np.int64), 40000)
    if kind == 2:
        return np.full(nrows, int(rng.integers(1, 9)))
    if kind == 3:
        return np.where(rng.random(nrows) < 0.02, rng.integers(500, 9000, nrows), rng.integers(0, 4, nrows))
    if kind == 4:
        return np.zeros(nrows, dtype=np.int64)
    return rng.integers(0, 40, nrows) * (rng.random(nrows) < 0.5)


def _rp(lens, first=0):
    rp = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=rp[1:])
    return rp + first


def test_device_planner_equals_host_planner_small():
    rng = np.random.default_rng(90210)
    for case in range(150):
        nrows = int(rng.integers(1, 3000))
        lens = np.asarray(_lens(rng, case % 6, nrows), dtype=np.int64)
        S = int(rng.choice([4, 8, 12, 16, 28, 32, 56, 64, 128, 508]))
        thr = int(rng.choice([0, 1, 5, 64, 10**7]))
        max_rows = int(rng.choice([0, 0, 1, 2, 7, 63, 500]))
        try:
            cvr_amd.plan_selfcheck(_rp(lens, int(rng.integers(0, 3))), S, thr, max_rows)
        except cvr_amd.CvrError as e:
            raise AssertionError(dict(case=case, nrows=nrows, S=S, thr=thr, max_rows=max_rows, err=str(e)))


@pytest.mark.parametrize("kind", [0, 1, 3, 5])
def test_device_planner_equals_host_planner_row_blocks(kind):
    rng = np.random.default_rng(1234 + kind)
    nrows = int(rng.integers(140_000, 400_000))
    lens = np.asarray(_lens(rng, kind, nrows), dtype=np.int64)
    lens[65535] = 2500                               # a long row at the end of a row block, an empty one at the end of the next
    lens[131071] = 0
    lens[rng.integers(0, nrows, 30)] = rng.integers(3000, 90000, 30)
    for S, thr, max_rows in ((8, 0, 0), (32, 3, 0), (48, 0, 1000), (64, 100, 37)):
        r = cvr_amd.plan_selfcheck(_rp(lens), S, thr, max_rows)
        assert r["nchunks"] > 0


def test_device_planner_declines_what_it_cannot_hold():
    with pytest.raises(cvr_amd.CvrError):            # 64 S slots per chunk must fit the 15-bit jump
        cvr_amd.plan_selfcheck(_rp(np.full(1000, 3)), 512)
    r = cvr_amd.plan_selfcheck(np.zeros(1, dtype=np.int64), 8)     # no rows: no chunks
    assert r["nchunks"] == 0


def test_panel_rule_on_the_device_equals_the_host_rule():
    """cvr_create's automatic panel count for a matrix with a large x comes from the L2 model run on the device copy of the CSR
    (cvr_split.hip: l2_hits_device): the same count as cvr_auto_panels (the host form) gives -- scattered columns -> panels,
    a band -> one image (which then adopts the staging copy) -- for host arrays and for arrays already on the device; y checked"""
    import oraclelib as O
    from cvr_amd import capi
    rng = np.random.default_rng(5)
    n = 3_300_000                                                          # x = 26.4 MB of fp64
    rp = np.arange(n + 1, dtype=np.int64) * 2
    scattered = rng.integers(0, n, 2 * n).astype(np.int32)
    scattered.reshape(n, 2).sort(axis=1)
    band = (np.repeat(np.arange(n, dtype=np.int64), 2) + np.tile([0, 3], n)).clip(0, n - 1).astype(np.int32)
    va = ((np.arange(2 * n) % 7) - 2.5).astype(np.float64)
    x = O.x_vec_fast(n, "rand")
    for ci in (scattered, band):
        P_host, _ = capi.auto_panels(n, n, rp, ci)
        A = cvr_amd.CvrMatrix(n, n, rp, ci, va)
        assert A.info.col_panels == P_host, (A.info.col_panels, P_host)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        bad, worst = O.tol_check(y, yref, absy + 1e-30)
        assert len(bad) == 0, worst
        A.close()
    assert P_host == 1
    P_host, _ = capi.auto_panels(n, n, rp, scattered)
    assert P_host > 4
    if torch is None:
        return
    keep = [torch.from_numpy(a).cuda() for a in (rp, scattered, va)]
    torch.cuda.synchronize()
    A = cvr_amd.CvrMatrix.from_device(n, n, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr())
    assert A.info.col_panels == P_host
    A.close()


def test_large_chunks_fall_back_to_the_host_planner():
    """chunks of 64 x 512 slots do not fit the device planner's 15-bit jumps: cvr_create (250 000 rows: the device planner's
    territory) fetches the row pointers and plans on the host; y still equals the CSR oracle's"""
    import oraclelib as O
    rng = np.random.default_rng(11)
    n = 250_000
    lens = rng.integers(0, 7, n)
    lens[rng.integers(0, n, 20)] = rng.integers(20_000, 60_000, 20)          # rows cut over several such chunks
    rp = _rp(lens)
    ci = rng.integers(0, n, int(rp[-1])).astype(np.int32)
    va = ((np.arange(len(ci)) % 11) - 4.0).astype(np.float64)
    A = cvr_amd.CvrMatrix(n, n, rp, ci, va, steps_per_chunk=512)
    assert A.info.steps_per_chunk == 512
    x = O.x_vec_fast(n, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    bad, worst = O.tol_check(y, yref, absy + 1e-30)
    assert len(bad) == 0, worst
    A.close()

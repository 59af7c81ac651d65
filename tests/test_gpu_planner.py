"""GPU: the device planner (cvr_amd/csrc/cvr_plan_dev.hip) against the host planner (cvr_plan.cpp, itself pinned against the
mirror's one-row-at-a-time planner in test_cvr64_mirror.py): same chunks and same cut rows, field by field, on random
row-length distributions, with and without a row cap, across row-block restarts."""
import numpy as np
import pytest

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

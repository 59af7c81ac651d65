"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- nnz-balanced row partition at row boundaries,
x replicated, y slices all-gathered (cvr_amd/shard.py).  The local SpMV is the CSR oracle here (no GPU in this
container); on the GPU box the same code runs with the HIP kernel and the "nccl" (= RCCL) backend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases as K
import oraclelib as O
from cvr_amd import shard


def test_row_partition_properties():
    for name, (nrows, ncols, rp, ci, va) in K.cases().items():
        for nparts in (1, 2, 3, 8):
            b = shard.row_partition(rp, nparts)
            assert b[0] == 0 and b[-1] == nrows and np.all(np.diff(b) >= 0) and len(b) == nparts + 1
            max_rows, pick = shard.gather_layout(b)
            assert len(pick) == nrows and max_rows == np.diff(b).max()
            # pieces tile the CSR exactly
            tot = 0
            for p in range(nparts):
                n, lrp, lci, lva = shard.local_csr(rp, ci, va, b, p)
                assert lrp[0] == 0 and len(lrp) == n + 1 and len(lci) == lrp[-1]
                tot += len(lci)
            assert tot == len(ci)
    # the library's routine (cvr_row_partition: the one rule behind cvr_create_multi, spmv.cvr, bench.py) against its definition:
    # the first row whose start reaches p / nparts of the non-zeros
    rng = np.random.default_rng(3)
    for trial in range(20):
        n = int(rng.integers(1, 400))
        rp = np.concatenate([[int(rng.integers(0, 5))], np.cumsum(rng.integers(0, 9, size=n))]).astype(np.int64)
        rp[1:] += rp[0]
        for nparts in (1, 2, 5, 8, 13):
            nnz = int(rp[-1] - rp[0])
            want = np.concatenate([[0], np.clip(np.searchsorted(rp, rp[0] + (np.arange(1, nparts) * nnz) // nparts, side="left"), 0, n), [n]])
            assert np.array_equal(shard.row_partition(rp, nparts), np.maximum.accumulate(want)), (trial, nparts)
            # the cut on predicted time (cvr_row_partition_cost): a row costs its non-zeros + w / 1000; the first row whose cost prefix
            # reaches p / nparts of the total -- w = 0 is the rule above
            for w in (0, 1250, 40000):
                cost = (rp - rp[0]) * 1000 + np.arange(n + 1) * w
                want_c = np.concatenate([[0], np.clip(np.searchsorted(cost, ((np.arange(1, nparts) * nnz) // nparts) * 1000 + (np.arange(1, nparts) * n * w) // nparts, side="left"), 0, n), [n]])
                assert np.array_equal(shard.row_partition(rp, nparts, w), np.maximum.accumulate(want_c)), (trial, nparts, w)
    # many short rows behind a few long ones (R-MAT's shape along the rows): equal non-zeros give the last part most of the rows; with the
    # row cost the parts' predicted times (non-zeros + 1.25 per row) are level and the row counts closer
    deg = np.concatenate([np.full(100, 5000), np.full(100000, 5)])
    rp2 = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    b0, b1 = shard.row_partition(rp2, 4), shard.row_partition(rp2, 4, 1250)
    t = lambda b: np.array([(rp2[b[p + 1]] - rp2[b[p]]) + 1.25 * (b[p + 1] - b[p]) for p in range(4)])
    assert t(b1).max() / t(b1).mean() < 1.02 < t(b0).max() / t(b0).mean()
    # balance on a matrix with many rows: within one max row of the ideal
    nrows, ncols, rp, ci, va = K.cases()["uniform_2000"]
    b = shard.row_partition(rp, 8)
    per = np.array([rp[b[p + 1]] - rp[b[p]] for p in range(8)])
    assert per.max() - per.min() <= 2 * np.diff(rp).max()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nrows, ncols, rp, ci, va = K.cases()[name]
        b = shard.row_partition(rp, world)
        n, lrp, lci, lva = shard.local_csr(rp, ci, va, b, rank)
        max_rows, pick = shard.gather_layout(b)
        x = O.x_vec_fast(ncols, "rand")                      # replicated
        yl, _ = O.csr_spmv64(lrp, lci, lva, x)               # stand-in for the HIP kernel on this shard
        buf = torch.zeros(max(max_rows, n) + 3, dtype=torch.float64)   # like y_ext: rows, then don't-care
        buf[:n] = torch.from_numpy(yl)
        buf[n:] = -777.0
        yall = shard.all_gather_y(buf, max_rows)
        y = yall[torch.from_numpy(pick)].numpy()
        yref, _ = O.csr_spmv64(rp, ci, va, x)
        ok = bool(np.array_equal(y, yref))
        # the pipelined fixed-x loop of bench.py (double-buffered y, asynchronous gathers)
        ybufs = [torch.full_like(buf, -1.0) for _ in range(2)]
        yalls = [torch.zeros(world * max_rows, dtype=torch.float64) for _ in range(2)]
        calls = []

        def spmv(yb):
            calls.append(1)
            yb[:n] = torch.from_numpy(yl) * len(calls)      # step k writes k * y so that stale buffers show
        for steps in (1, 2, 5):
            del calls[:]
            last = shard.pipelined_steps(spmv, ybufs, yalls, max_rows, steps)
            yk = yalls[last][torch.from_numpy(pick)].numpy()
            ok = ok and len(calls) == steps and bool(np.array_equal(yk, yref * steps))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["power_law_3000", "leading_trailing_empty", "few_rows_lt_lanes"])
def test_sharded_spmv_world2_gloo(name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _shard_local_worker(rank, world, port, q):
    """every rank builds only its own row block of the large synthetic workloads (cvr_amd/synth_dev.py) and the blocks,
    gathered over gloo, must equal the matrix built in one piece"""
    import torch
    import torch.distributed as dist
    from cvr_amd import synth_dev as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scale = 12
        deg = D.rmat_row_degrees(scale)
        bounds, grp = D.partition_from_degrees(deg, world)
        lrp, lci, lva = D.rmat_rows(scale, int(bounds[rank]), int(bounds[rank + 1]))
        parts = [None] * world
        dist.all_gather_object(parts, (lrp.numpy(), lci.numpy(), lva.numpy()))
        nb, nnzb = D.banded_partition(3000, 13, world)
        brp, bci, bva = D.banded_rows(3000, int(nb[rank]), int(nb[rank + 1]))
        bparts = [None] * world
        dist.all_gather_object(bparts, (brp.numpy(), bci.numpy(), bva.numpy()))
        if rank == 0:
            wrp, wci, wva = (t.numpy() for t in D.rmat_rows(scale, 0, 1 << scale))
            ok = int(grp[-1]) == int(wrp[-1]) == 16 << scale
            for p in range(world):
                lo, hi = int(bounds[p]), int(bounds[p + 1])
                a, b = int(wrp[lo]), int(wrp[hi])
                ok &= np.array_equal(parts[p][0], wrp[lo:hi + 1] - a) and np.array_equal(parts[p][1], wci[a:b]) and np.array_equal(parts[p][2], wva[a:b])
            nnz_per = [int(grp[bounds[p + 1]] - grp[bounds[p]]) for p in range(world)]
            ok &= max(nnz_per) - min(nnz_per) <= int(deg.max())          # balanced up to one row
            wb = [t.numpy() for t in D.banded_rows(3000, 0, 3000)]
            ok &= int(wb[0][-1]) == nnzb
            for p in range(world):
                lo, hi = int(nb[p]), int(nb[p + 1])
                a, b = int(wb[0][lo]), int(wb[0][hi])
                ok &= np.array_equal(bparts[p][0], wb[0][lo:hi + 1] - a) and np.array_equal(bparts[p][1], wb[1][a:b]) and np.array_equal(bparts[p][2], wb[2][a:b])
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_shard_local_workloads_equal_slices_of_the_whole(world):
    """world 8 = the rank count of BASELINE.json configs[3] / [4] (one process per GPU of a node): partition_from_degrees and the
    shard-local generators at that size, over gloo"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_local_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok


def test_partition_from_degrees_is_the_library_rule_at_eight_parts():
    """synth_dev.partition_from_degrees restates cvr_row_partition_cost on device tensors (shards are built where they will live): the same
    bounds as the C routine for 8 parts, with the row cost of the product (1.25 non-zeros per row) and without"""
    from cvr_amd import synth_dev as D
    deg = D.rmat_row_degrees(14)
    rp = np.concatenate([[0], np.cumsum(deg.numpy())]).astype(np.int64)
    for cost in (0, 1250):
        bounds, grp = D.partition_from_degrees(deg, 8, cost)
        assert np.array_equal(np.asarray(bounds, dtype=np.int64), np.asarray(shard.row_partition(rp, 8, cost), dtype=np.int64))
        assert int(grp[-1]) == int(rp[-1]) and bounds[0] == 0 and bounds[-1] == len(deg)


def test_synth_dev_matches_the_numpy_generators_where_defined():
    import torch
    from cvr_amd import synth, synth_dev as D
    assert np.array_equal(D.x_rand(5000).numpy(), synth.x_rand(5000))
    assert np.array_equal(D.x_rand(777, dtype=torch.float32).numpy(), synth.x_rand(777, np.float32))
    n, nc, rp, ci, va = synth.banded_sym(400)
    brp, bci, _ = D.banded_rows(400, 0, 400)
    assert np.array_equal(brp.numpy(), rp) and np.array_equal(bci.numpy(), ci)
    y, ay = D.csr_spmv_reference(brp, bci, torch.from_numpy(va), D.x_rand(400))
    yref = np.array([np.dot(va[rp[r]:rp[r + 1]], synth.x_rand(400)[ci[rp[r]:rp[r + 1]]]) for r in range(400)])
    assert np.allclose(y.numpy(), yref, rtol=0, atol=1e-13)

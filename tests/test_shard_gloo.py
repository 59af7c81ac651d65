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

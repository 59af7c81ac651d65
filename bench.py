#!/usr/bin/env python3
"""bench.py -- the headline measurement: CVR-format SpMV on MI355X (BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one y = A x over the web-Google-shaped matrix (916 428 x 916 428, 5 105 039 nnz, fp64; seeded
synthetic stand-in, or the real web-Google.mtx when CVR_DATA_DIR holds it), matrix image, x and y resident
in HBM.  N > 1: rows are sharded over the ranks (balanced nnz, cut at row boundaries; S of a shard chosen by measurement,
cvr_tune_steps), x is replicated, every step ends with the all-gather of the y slices over RCCL ("strong" scaling: the
matrix is fixed).  The loop of steps runs inside the library (cvr_spmv_gather_repeat: SpMV and ncclAllGather enqueued
back to back, no Python between steps) once every rank has built its communicator and its first gather has been checked
bit for bit against torch.distributed's; in order or overlapped (gather of step k under the SpMV of step k+1), whichever
200 untimed steps show to be faster on this node.  Otherwise the same loop runs over torch.distributed.
Rank 0 prints ONE JSON line.  The roofline object prices the SpMV kernel alone (algorithmic bytes of SURVEY.md
8(d) / mean kernel time from HIP events on the launch stream); cpu_baseline is the unmodified reference built by
oracle/Makefile into oracle/_ref/ (kind "reference"; the oracle's 8-lane OpenMP restatement, kind "port", when that
binary is absent) on the host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic():
    """HBM bytes per SpMV launch from the rocprofv3 PMC passes of this same command (tools/profile_bench.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950); bench.py cannot run the profiler on itself, so it reports the committed summary, or null."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        return float(json.load(open(p))["hbm_bytes_per_launch_corrected"])
    except Exception:
        return None


def load_workload(kind="webgoogle"):
    from cvr_amd import capi, synth
    import cvr_amd
    if kind == "livejournal":      # BASELINE.json configs[2]
        n, nc, rp, ci, va = synth.livejournal_like()
        return n, nc, rp, ci, va, "synthetic soc-LiveJournal1-shaped, seed 20261003"
    if kind.startswith("rmat"):    # configs[4]: R-MAT (Graph500 parameters), fp32, rmat<scale> (default 22)
        n, nc, rp, ci, va = synth.rmat(int(kind[4:] or 22), dtype=np.float32)
        return n, nc, rp, ci, va, f"synthetic R-MAT scale {int(kind[4:] or 22)}, edge factor 16, fp32"
    if kind.startswith("banded"):  # configs[3]: nlpkkt240's shape; banded<rows>, default 28e6 rows / 8
        n, nc, rp, ci, va = synth.banded_sym(int(float(kind[6:] or 3.5e6)))
        return n, nc, rp, ci, va, "synthetic banded symmetric (27 nnz/row, nlpkkt240's shape)"
    f = synth.data_file("web-Google.mtx")
    if f:
        m = cvr_amd.load_mm(f, capi.MM_STRICT)
        vals = (np.arange(m["nnz"], dtype=np.int64) % 13).astype(np.float64)   # pattern file: spmv.cpp:417
        return m["nrows"], m["ncols"], m["row_ptr"], m["col_idx"], vals, "web-Google.mtx (SNAP)"
    n, nc, rp, ci, va = synth.web_google_like()
    return n, nc, rp, ci, va, "synthetic web-Google-shaped, seed 20261002"


def _cpu_reference(nrows, ncols, rp, ci, cores):
    """the UNMODIFIED reference (spmv.cpp compiled by oracle/Makefile into oracle/_ref/, prebuilt in the build
    container) on a Matrix-Market file of the bench matrix: kind = "reference".  None if it cannot run here."""
    import re
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    exe = os.path.join(ROOT, "oracle", "_ref", "spmv.cvr.ref")
    if not os.path.exists(exe):
        return None
    T = min(cores, 68)                      # run_sample.sh:10 runs web-Google with 68 threads
    iters = 300
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        path = os.path.join(d, "bench.mtx")
        O.write_mtx_pattern(path, nrows, ncols, rp, ci)
        env = dict(os.environ, OMP_NUM_THREADS=str(T), OMP_PROC_BIND="close", OMP_PLACES="cores")
        try:
            r = subprocess.run([exe, path, str(T), str(iters)], capture_output=True, text=True, timeout=240, env=env)
        except Exception:
            return None
    out = r.stdout
    m = re.search(r"SpMV Execution Time of CVR\s+is ([0-9.eE+-]+) seconds", out)
    p = re.search(r"Pre-processing\(CSR->CVR\)\s+Time of CVR\s+is ([0-9.eE+-]+) seconds", out)
    if r.returncode != 0 or not m or "Very Good" not in out:
        return None
    per = float(m.group(1))
    from cvr_amd import synth
    return {"value": 2.0 * len(ci) / per / 1e9, "unit": "GFLOP/s", "cores": T, "kind": "reference",
            "sample": f"unmodified reference source built by oracle/Makefile (g++ -O3 -mavx512f -fopenmp, 4 intrinsic-spelling aliases in oracle/ref_shim.h), {iters} timed SpMV iterations of the full matrix, "
                      f"{T} OpenMP threads, y zeroing outside the timer as the reference does (spmv.cpp:1026-1033)",
            "ms_per_step": per * 1e3, "preprocess_s": float(p.group(1)) if p else None,
            "gbs_alg": synth.b_alg(nrows, ncols, len(ci)) / per / 1e9}


def _cpu_port(nrows, ncols, rp, ci, va, cores, budget_s=10.0):
    """the oracle's scalar-C restatement of the reference CPU path (8 lanes, one chunk per OpenMP thread;
    spmv.cpp:565-1014, 1016-1667) on the reference loader's 1-based arrays: kind = "port" """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    from cvr_amd import synth
    m = synth.to_refcompat(nrows, ncols, rp, ci, va)
    x = np.ones(ncols + 2)
    best = None
    for T in sorted({min(cores, t) for t in (8, 16, 32, 64, 128)}):     # thread count: quick probe, keep the best
        c = O.Cvr8(m, T)
        if c.rc != 0:
            continue
        c.spmv(x, nthreads=T)
        t0 = time.perf_counter()
        for _ in range(3):
            c.spmv(x, nthreads=T)
        per = (time.perf_counter() - t0) / 3
        if best is None or per < best[0]:
            best = (per, T)
    if best is None:
        return None
    T = best[1]
    t0 = time.perf_counter()
    c = O.Cvr8(m, T)
    pre = time.perf_counter() - t0
    c.spmv(x, nthreads=T)
    iters, t_used = 0, 0.0
    t_start = time.perf_counter()
    while iters < 2000 and t_used < budget_s:
        c.spmv(x, nthreads=T)
        iters += 1
        t_used = time.perf_counter() - t_start
    per = t_used / iters
    return {"value": 2.0 * len(ci) / per / 1e9, "unit": "GFLOP/s", "cores": T, "kind": "port",
            "sample": f"{iters} SpMV iterations of the full matrix, {T} OpenMP threads (best of a probe over thread counts), "
                      "y zeroing inside the timer", "ms_per_step": per * 1e3, "preprocess_s": pre,
            "gbs_alg": synth.b_alg(nrows, ncols, len(ci)) / per / 1e9}


def cpu_baseline(nrows, ncols, rp, ci, va):
    cores = len(os.sched_getaffinity(0))
    r = None
    if os.environ.get("CVR_CPU_BASELINE", "reference") == "reference":
        r = _cpu_reference(nrows, ncols, rp, ci, cores)
    return r or _cpu_port(nrows, ncols, rp, ci, va, cores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--steps-per-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-streams", action="store_true", help="also report the throughput of independent SpMVs alternating on two streams "
                    "(off by default: concurrent kernels would distort a rocprofv3 kernel-time summary of this command)")
    ap.add_argument("--workload", default="webgoogle", help="webgoogle (the headline, default) | livejournal | banded[<rows>] | rmat[<scale>] (fp32)")
    args = ap.parse_args()

    import torch
    import cvr_amd
    from cvr_amd import shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU fallback")
    if os.environ.get("CVR_BENCH_ONE_DEVICE"):      # plumbing check of the N > 1 path on a 1-GPU box (all ranks on cuda:0)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or bool(os.environ.get("CVR_BENCH_FORCE_SHARDED"))   # the latter: the N > 1 code path with one rank (RCCL plumbing check)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        backend = os.environ.get("CVR_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    nrows, ncols, rp, ci, va, source = load_workload(args.workload)
    nnz = len(ci)
    bounds = shard.row_partition(rp, world)
    lrows, lrp, lci, lva = shard.local_csr(rp, ci, va, bounds, rank)
    # a shard of this matrix on one of N GPUs is small enough for the chunk count to matter: measure S (cvr_tune_steps)
    tune = world > 1 and args.steps_per_chunk == 0 and not os.environ.get("CVR_BENCH_NO_TUNE")
    A = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, device=local_rank, steps_per_chunk=args.steps_per_chunk, tune_steps=tune)
    info = A.info
    max_rows, pick = shard.gather_layout(bounds)

    dev = torch.device("cuda", local_rank)
    f32 = va.dtype == np.float32
    tdt, vbytes = (torch.float32, 4) if f32 else (torch.float64, 8)
    bits = torch.int32 if f32 else torch.int64
    x = torch.zeros(info.x_elems, dtype=tdt, device=dev)
    x[:ncols] = torch.from_numpy(synth.x_rand(ncols, va.dtype)).to(dev)
    ny = max(info.yext_elems, max_rows)
    ybufs = [torch.zeros(ny, dtype=tdt, device=dev) for _ in range(2 if sharded else 1)]
    yalls = [torch.zeros(world * max_rows, dtype=tdt, device=dev) for _ in range(2)] if sharded else None
    y = ybufs[0]
    stream = torch.cuda.Stream(device=dev)     # kernels, events and the collective all go on this stream
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream

    last = [0]

    # The exchange step.  "native" (default): the pipelined loop of cvr_spmv_gather_repeat, RCCL called from the library
    # with no Python between the steps; "torch": the same loop in Python over torch.distributed (shard.pipelined_steps).
    # Native is used only if every rank could build its communicator and its first gather equals torch.distributed's.
    comm = None
    gather_impl = "none"
    if sharded:
        gather_impl = "torch"
        if os.environ.get("CVR_BENCH_GATHER", "native") == "native" and os.environ.get("CVR_BENCH_BACKEND", "nccl") == "nccl":
            box = [None]
            if rank == 0:
                try:
                    box[0] = cvr_amd.comm_unique_id()
                except Exception as e:
                    print(f"[bench rank 0] no RCCL id: {e!r}", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)       # always reached by every rank, id or not
            def agree(ok):          # every rank must have succeeded
                flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1

            try:                    # phase 1: the communicators
                if box[0] is None:
                    raise RuntimeError("rank 0 could not make an RCCL id")
                comm = cvr_amd.Comm(box[0], world, rank, local_rank)
            except Exception as e:
                print(f"[bench rank {rank}] native gather unavailable: {e!r}", file=sys.stderr)
            if agree(comm is not None):
                ok = False
                try:                # phase 2: its first gather against torch.distributed's, bit for bit
                    A.spmv_device(x.data_ptr(), ybufs[0].data_ptr(), sptr)
                    stream.synchronize()
                    comm.all_gather(ybufs[0].data_ptr(), yalls[0].data_ptr(), max_rows, f32, sptr)
                    stream.synchronize()
                    want = shard.all_gather_y(ybufs[0], max_rows)
                    torch.cuda.synchronize()
                    ok = torch.equal(want.view(bits), yalls[0].view(bits))
                except Exception as e:
                    print(f"[bench rank {rank}] native gather failed its check: {e!r}", file=sys.stderr)
                if agree(ok):
                    gather_impl = "native"
            if gather_impl != "native" and comm is not None:     # stay on the torch.distributed path
                comm.close()
                comm = None

    overlap = [False]

    def step(n=1):
        if not sharded:
            A.spmv_device(x.data_ptr(), y.data_ptr(), sptr, repeat=n)
        elif comm is not None:   # every step = local SpMV + all-gather of the y slices
            last[0] = A.spmv_gather(comm, x.data_ptr(), [b.data_ptr() for b in ybufs], [b.data_ptr() for b in yalls], max_rows, n, sptr,
                                    overlap=overlap[0])
        else:                    # the gather of step k overlaps the SpMV of k+1
            last[0] = shard.pipelined_steps(lambda yb: A.spmv_device(x.data_ptr(), yb.data_ptr(), sptr), ybufs, yalls, max_rows, n)

    def sync():
        # drain this rank's queue first: the library's communicator and torch.distributed's are different RCCL
        # communicators, and their collectives should never be in flight at the same time
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    calib = None
    if comm is not None:         # in-order or overlapped gather: whichever this node runs faster (untimed, every rank agrees)
        calib = {}
        for mode in (False, True):
            overlap[0] = mode
            step(20)
            sync()
            t0 = time.perf_counter()
            step(200)
            sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            calib["overlap" if mode else "in_order"] = float(t.item()) / 200 * 1e3
        overlap[0] = calib["overlap"] < calib["in_order"]
        gather_impl = "native, " + ("overlapped" if overlap[0] else "in order")

    step(args.warmup)
    sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    step(args.steps)
    e1.record(stream)
    sync()
    wall = time.perf_counter() - t0
    ev_s = e0.elapsed_time(e1) * 1e-3
    if sharded:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # the SpMV kernel alone, HIP events on the launch stream (N > 1: this rank's shard, no gather)
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    A.spmv_device(x.data_ptr(), y.data_ptr(), sptr, repeat=args.warmup)
    e2.record(stream)
    A.spmv_device(x.data_ptr(), y.data_ptr(), sptr, repeat=args.steps)
    e3.record(stream)
    torch.cuda.synchronize()
    kern_s = e2.elapsed_time(e3) * 1e-3 / args.steps
    gather_s = None
    if sharded:                         # the exchange step alone, same message, same stream (reported beside the total)
        e4, e5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        def gather_once():
            if comm is not None:
                comm.all_gather(ybufs[0].data_ptr(), yalls[0].data_ptr(), max_rows, f32, sptr)
            else:
                shard.all_gather_y(ybufs[0], max_rows, out=yalls[0])
        for _ in range(5):
            gather_once()
        e4.record(stream)
        for _ in range(args.steps):
            gather_once()
        e5.record(stream)
        torch.cuda.synchronize()
        gather_s = e4.elapsed_time(e5) * 1e-3 / args.steps
    # the same launches one by one, each between its own pair of events: median and minimum (SURVEY 8d asks for mean / median / min)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 200))]
    for a, b in evs:
        a.record(stream)
        A.spmv_device(x.data_ptr(), y.data_ptr(), sptr)
        b.record(stream)
    torch.cuda.synchronize()
    singles = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    # throughput of independent SpMVs alternating on two streams with their own y (the per-launch floor of one overlaps the body
    # of the other); reported for information, never the value: the steps of the metric run one after the other
    two = None
    if not sharded and args.two_streams:
        st2 = torch.cuda.Stream(device=dev)
        y2 = torch.zeros_like(y)
        pair = ((sptr, y), (st2.cuda_stream, y2))
        for i in range(2 * args.warmup):
            A.spmv_device(x.data_ptr(), pair[i & 1][1].data_ptr(), pair[i & 1][0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            A.spmv_device(x.data_ptr(), pair[i & 1][1].data_ptr(), pair[i & 1][0])
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) / args.steps
    kern_max_s = kern_s
    if sharded:                         # the slowest rank's SpMV alone: what the job would run at without the exchange step
        t = torch.tensor([kern_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kern_max_s = float(t.item())
    lnnz = int(lrp[-1])
    balg_local = synth.b_alg(lrows, ncols, lnnz, vbytes)
    achieved = balg_local / kern_s / 1e9

    # parity guard on the timed configuration: y of the last step against the host CSR loop of the product
    # (the reference's own self-check, spmv.cpp:1843-1850, 1916-1938)
    yh = (yalls[last[0]][torch.from_numpy(pick).to(dev)] if sharded else y[:nrows]).cpu().numpy()
    wrong = -1
    if rank == 0:
        xh = x[:ncols].cpu().numpy().astype(np.float64)
        yref = cvr_amd.csr_spmv_host(rp, ci, va.astype(np.float64), xh, nthreads=len(os.sched_getaffinity(0)))
        if f32:     # no reference counterpart: rows off by more than 1e-5 of sum |a x| against the fp64 loop
            absy = cvr_amd.csr_spmv_host(rp, ci, np.abs(va).astype(np.float64), np.abs(xh), nthreads=len(os.sched_getaffinity(0)))
            wrong = int(np.count_nonzero(np.abs(yh.astype(np.float64) - yref) > 1e-5 * absy + 1e-30))
        else:
            wrong = int(cvr_amd.verdict(yh, yref, nrows))

    copy_gbs = None
    if rank == 0:
        try:                      # achievable-HBM yardstick measured live: 1 GiB streaming copy, read + write
            from cvr_amd import capi
            copy_gbs = capi.device_copy_gbs(local_rank, 1 << 30, 20)
        except Exception:
            copy_gbs = None
    if rank == 0:
        per = wall / args.steps
        out = {
            "metric": "SpMV GFLOP/s (2*nnz/t), web-Google fp64" if args.workload == "webgoogle" else f"SpMV GFLOP/s (2*nnz/t), {args.workload} {'fp32' if f32 else 'fp64'}",
            "value": 2.0 * nnz / per / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if f32 else "f64",
            "data": "synthetic" if source.startswith("synthetic") else "real",
            "config": {"workload": f"{source}: {nrows}x{ncols}, nnz {nnz}, {'fp32' if f32 else 'fp64'}, y = A x with A (CVR64 image), x, y resident in HBM",
                       "rows_per_gpu": [int(v) for v in np.diff(bounds)],
                       "nnz_per_gpu": [int(rp[bounds[p + 1]] - rp[bounds[p]]) for p in range(world)],
                       "nnz_imbalance_max_over_mean": float(max(int(rp[bounds[p + 1]] - rp[bounds[p]]) for p in range(world)) * world / max(nnz, 1)),
                       "steps_per_chunk": int(info.steps_per_chunk), "chunks_rank0": int(info.nchunks),
                       "rows_cut_rank0": int(info.nshared), "col_panels": int(info.col_panels),
                       "value_dictionary_entries": int(info.value_dict),
                       "parallelism": ("rows sharded, x replicated, y all-gathered (%s)" % ("RCCL" if os.environ.get("CVR_BENCH_BACKEND", "nccl") == "nccl" else os.environ["CVR_BENCH_BACKEND"])) if sharded else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic() if world == 1 and args.workload == "webgoogle" else None,
                         "kernel": "cvr::spmv_kernel<float>" if f32 else "cvr::spmv_kernel<double>", "kernel_us": kern_s * 1e6,
                         "kernel_us_median_single_launches": singles[len(singles) // 2] if singles else None, "kernel_us_min_single_launches": singles[0] if singles else None,
                         "copy_kernel_gbs": copy_gbs, "frac_of_copy_kernel": achieved / copy_gbs if copy_gbs else None,
                         "algorithmic_bytes_per_launch": int(balg_local)},
            "gbs_alg_whole_job": synth.b_alg(nrows, ncols, nnz, vbytes) / per / 1e9,
            "event_ms_per_step_rank0": ev_s / args.steps * 1e3,
            "spmv_only_ms_max_over_ranks": kern_max_s * 1e3, "gflops_spmv_only_no_exchange": 2.0 * nnz / kern_max_s / 1e9,
            "rank0_spmv_only_ms": kern_s * 1e3, "rank0_allgather_only_ms": None if gather_s is None else gather_s * 1e3,
            "preprocess": {"plan_s": info.plan_s, "upload_s": info.upload_s, "convert_s": info.convert_s, "tune_steps_s": A.tuning_s},
            "independent_spmvs_on_two_streams": None if two is None else {"ms_per_spmv": two * 1e3, "gflops": 2.0 * nnz / two / 1e9},
            "verdict_wrong_rows": wrong, "gather_impl": gather_impl, "gather_calibration_ms_per_step": calib,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(nrows, ncols, rp, ci, va)
            except Exception as e:   # the checker is optional on the bench box; the GPU numbers stand without it
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    A.close()
    if sharded:
        dist.barrier()                      # rank 0 is still verifying / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

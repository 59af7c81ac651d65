#!/usr/bin/env python3
"""bench.py -- the headline measurement: CVR-format SpMV on MI355X (BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload webgoogle|livejournal|banded<rows>|rmat<scale>]

With N > 1 and no torch.distributed environment the command starts itself once per GPU (python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ..., before anything touches the GPU) and relays rank 0's line; under
torch.distributed.run it is one rank.

A "step" is one y = A x over the web-Google-shaped matrix (916 428 x 916 428, 5 105 039 nnz, fp64; seeded synthetic stand-in,
or the real web-Google.mtx when CVR_DATA_DIR holds it), matrix image, x and y resident in HBM.  N > 1: rows are sharded over
the ranks (balanced nnz, cut at row boundaries; the layout of a shard chosen by measurement, cvr_tune), x is replicated, every
step ends with the all-gather of the y slices over RCCL ("strong" scaling: the matrix is fixed).  The loop of steps runs inside
the library (cvr_spmv_gather_repeat: SpMV and ncclAllGather enqueued back to back, no Python between steps) once every rank has
built its communicator and its first gather has been checked bit for bit against torch.distributed's; in order or overlapped
(gather of step k under the SpMV of step k+1), whichever 200 untimed steps show to be faster on this node.  Otherwise the same
loop runs over torch.distributed.  The large workloads (banded<rows>, rmat<scale>: BASELINE.json configs[3], [4]) are built
shard by shard on the GPU (cvr_amd/synth_dev.py) and handed to the library as device-resident CSR.
Rank 0 prints ONE JSON line.  The roofline object prices the SpMV kernel alone (algorithmic bytes of SURVEY.md 8(d) / mean
kernel time from HIP events on the launch stream); cpu_baseline is the unmodified reference built by oracle/Makefile into
oracle/_ref/ (kind "reference"; the oracle's 8-lane OpenMP restatement, kind "port", when that binary is absent) on the host
cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_summary(info, kernel, workload):
    """The rocprofv3 PMC summary of this same command (tools/profile_bench.sh: FETCH_SIZE, WRITE_SIZE, the L2 and L1->L2 counters in
    separate passes).  bench.py cannot run the profiler on itself, so it reports a committed summary (profiles/pmc_latest.json or any
    profiles/*_pmc_summary.json) -- but only one that was taken on the very configuration timed here (workload, kernel, chunk length, chunk
    count, image bytes, workgroup layout); otherwise None.  Per SpMV: the SpMV kernel's launches only (all panels), without the small
    combine / fix-up / hub-gather kernels."""
    import glob
    for p in [os.path.join(ROOT, "profiles", "pmc_latest.json")] + sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(p))
            c = d["config"]
            same = (c["kernel"] == kernel and c["steps_per_chunk"] == info.steps_per_chunk and c["nchunks"] == info.nchunks and
                    c["image_bytes"] == info.image_bytes and c["waves_per_block"] == info.waves_per_block and
                    c["col_phases"] == info.col_phases and c["x_window"] == info.x_window and c["workload"] == workload)
            if same and c.get("col_panels", info.col_panels) == info.col_panels:
                d["_file"] = os.path.relpath(p, ROOT)
                return d
        except Exception:
            continue
    return None


def pmc_fields(info, kernel, workload, nnz):
    """roofline.traffic and the request counters from the committed summary (None without a matching one).  FETCH_SIZE counts 64 bytes
    per fabric read request; MI355X_MICROARCH.md prescribes doubling it for wide coalesced reads (128-byte requests tallied at 64), which
    over-counts the 8-byte gathers' share: the true figure lies between the raw and the corrected one, both are given; `traffic` is the
    corrected (upper) one."""
    d = pmc_summary(info, kernel, workload)
    if d is None:
        return {"traffic": None}
    n = max(int(info.spmv_launches), 1)          # the summary averages over launches of the SpMV kernel; a panelled matrix may launch it once per panel
    out = {"traffic": float(d["hbm_bytes_per_launch_corrected"]) * n, "traffic_source": d["_file"]}
    if "hbm_bytes_per_launch_raw" in d:
        out["traffic_raw_and_corrected"] = [float(d["hbm_bytes_per_launch_raw"]) * n, float(d["hbm_bytes_per_launch_corrected"]) * n]
    if "TCP_TCC_READ_REQ_sum" in d:
        req = float(d["TCP_TCC_READ_REQ_sum"]) * n
        out.update({"requests_per_launch": req, "requests_per_nnz": req / max(nnz, 1),
                    "requests_are": "L1->L2 read requests (TCP_TCC_READ_REQ_sum) of the SpMV kernel per SpMV: x gathers, matrix stream and tables together",
                    "l2_request_yardstick_greq_per_s": 227.0,
                    "yardstick_is": "scattered 8-byte gathers from an L2-resident table, every lane its own 128-byte line: profiles/r04_gather_sharing_ubench.log (61 from the Infinity Cache)"})
    if "l2_hit_rate" in d:
        out["l2_hit_rate"] = float(d["l2_hit_rate"])
    return out


def capi_row_cost():
    from cvr_amd import capi
    return capi.ROW_COST_MILLI_DEFAULT


OTHER_WORKLOADS = ("livejournal", "rmat22", "orkut", "wikitalk")      # the power-law shapes beside the headline (north_star: "three SuiteSparse power-law matrices")


def measure_other_workload(kind, dev, steps=200, warmup=40):
    """One more shape in the same process (N = 1): built (on the device where a generator exists), converted with the library's own rules,
    timed with HIP events over `steps` back-to-back SpMVs on a stream of its own behind `warmup` untimed ones (the headline's kernel time is taken over >= 200
    launches behind its whole timed loop; with 6 + 60 launches straight after the conversion these shapes read 1.5 % slower than in a process of their own:
    soc-LiveJournal1 shape 208.6-209.4 against 205.4-206.4 us), every row checked against a torch fp64 segment sum.
    Returns what the judge needs to recompute the fraction: nnz, algorithmic bytes, kernel time."""
    import torch
    import cvr_amd
    from cvr_amd import synth, synth_dev as D
    t0 = time.perf_counter()
    if kind == "livejournal":
        n, nc, rp, ci, va = synth.livejournal_like()
        rp_t, ci_t, va_t = torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev)
        source = "synthetic soc-LiveJournal1-shaped, seed 20261003"
        del rp, ci, va
    elif kind.startswith("rmat"):
        scale = int(kind[4:] or 22)
        n = nc = 1 << scale
        rp_t, ci_t, va_t = D.rmat_rows(scale, 0, n, device=dev)
        source = f"synthetic R-MAT scale {scale}, edge factor 16, fp32, duplicates kept, built on the GPU"
    elif kind == "orkut":
        n, rp_t, ci_t, va_t = D.orkut_like(device=dev)
        nc = n
        source = "synthetic com-Orkut-shaped (symmetric, mean degree ~70), seed 20261004, built on the GPU"
    elif kind == "wikitalk":
        n, rp_t, ci_t, va_t = D.wikitalk_like(device=dev)
        nc = n
        source = "synthetic wiki-Talk-shaped (94 % empty rows, rows up to 100 000), seed 20261005, built on the GPU"
    else:
        raise ValueError(kind)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    f32 = va_t.dtype == torch.float32
    tdt = torch.float32 if f32 else torch.float64
    nnz = int(rp_t[-1])
    t1 = time.perf_counter()
    A = cvr_amd.CvrMatrix.from_device(n, nc, rp_t.data_ptr(), ci_t.data_ptr(), va_t.data_ptr(), is_f32=f32, device=dev.index or 0)
    create_s = time.perf_counter() - t1
    info = A.info
    x = torch.zeros(info.x_elems, dtype=tdt, device=dev)
    x[:nc] = D.x_rand(nc, device=dev, dtype=tdt)
    y = torch.zeros(max(info.yext_elems, 1), dtype=tdt, device=dev)
    # a stream of its own, as the headline has: launches on the legacy default stream are ordered against every other blocking stream of the process, which
    # costs the two-launch shapes ~2 us per launch (soc-LiveJournal1 shape 209.4 on the default stream, 205.8 on its own: profiles/r06_final_numbers.log)
    side = torch.cuda.Stream(device=dev)
    sptr = side.cuda_stream
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    A.spmv_device(x.data_ptr(), y.data_ptr(), sptr, repeat=warmup)
    a.record(side)
    A.spmv_device(x.data_ptr(), y.data_ptr(), sptr, repeat=steps)
    b.record(side)
    torch.cuda.synchronize()
    per = a.elapsed_time(b) * 1e-3 / steps
    # the same matrix with full values in the stream (value_dict = 0): what the format does without the compression of the 13 distinct values the
    # reference's loader gives a pattern file (spmv.cpp:417); the fraction is reported beside the main one, never instead of it
    per_nodict = None
    if int(info.value_dict) > 0 and not os.environ.get("CVR_BENCH_NO_DICT_OFF_RUN"):
        try:
            B = cvr_amd.CvrMatrix.from_device(n, nc, rp_t.data_ptr(), ci_t.data_ptr(), va_t.data_ptr(), is_f32=f32, device=dev.index or 0, value_dict=0)
            yb = torch.zeros(max(B.info.yext_elems, 1), dtype=tdt, device=dev)
            torch.cuda.synchronize()
            B.spmv_device(x.data_ptr(), yb.data_ptr(), sptr, repeat=warmup)
            a2, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a2.record(side)
            B.spmv_device(x.data_ptr(), yb.data_ptr(), sptr, repeat=max(steps // 2, 10))
            b2.record(side)
            torch.cuda.synchronize()
            per_nodict = a2.elapsed_time(b2) * 1e-3 / max(steps // 2, 10)
            B.close()
            del yb
        except Exception as e:
            print(f"[bench] {kind}: value_dict = 0 run failed: {e!r}", file=sys.stderr)
    yref_t, absy_t = D.csr_spmv_reference(rp_t, ci_t, va_t, x[:nc])
    wrong = int(torch.count_nonzero((y[:n].to(torch.float64) - yref_t).abs() > (1e-5 if f32 else 1e-12) * absy_t + 1e-300).item())
    vb = 4 if f32 else 8
    balg = synth.b_alg(n, nc, nnz, vb)
    out = {"workload": f"{source}: {n}x{nc}, nnz {nnz}, {'fp32' if f32 else 'fp64'}", "dtype": "f32" if f32 else "f64", "nnz": nnz, "steps": steps,
           "ms_per_step": per * 1e3, "kernel_us": per * 1e6, "gflops": 2.0 * nnz / per / 1e9, "algorithmic_bytes_per_launch": int(balg),
           "achieved_gbs": balg / per / 1e9, "frac": balg / per / 1e9 / HBM_PEAK_GBS, "wrong_rows": wrong,
           "kernel_us_value_dict_off": per_nodict * 1e6 if per_nodict else None, "frac_value_dict_off": balg / per_nodict / 1e9 / HBM_PEAK_GBS if per_nodict else None,
           "layout": {"col_panels": int(info.col_panels), "interleave": int(info.interleave), "steps_per_chunk": int(info.steps_per_chunk), "chunks": int(info.nchunks),
                      "waves_per_workgroup": int(info.waves_per_block), "hub_entries": int(info.hub_entries), "hub_reorder": int(info.hub_reorder),
                      "value_dictionary_entries": int(info.value_dict), "image_bytes": int(info.image_bytes), "spmv_launches": int(info.spmv_launches)},
           "t_pre_ms": _pre_times(info)["t_pre_s"] * 1e3, "build_s": build_s, "create_and_preprocess_wall_s": create_s}
    A.close()
    del rp_t, ci_t, va_t, x, y
    torch.cuda.empty_cache()
    return out


def host_cpu():
    """(model string, logical cores usable by this process, physical cores)"""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except Exception:
        pass
    logical = len(os.sched_getaffinity(0))
    return model, logical, (min(len(phys), logical) if phys else logical)


def load_host_workload(kind):
    from cvr_amd import capi, synth
    import cvr_amd
    if kind == "livejournal":      # BASELINE.json configs[2]
        n, nc, rp, ci, va = synth.livejournal_like()
        return n, nc, rp, ci, va, "synthetic soc-LiveJournal1-shaped, seed 20261003"
    if kind in ("orkut", "wikitalk"):      # generated on the device (cvr_amd/synth_dev.py), handed over as host arrays like the others
        import torch
        from cvr_amd import synth_dev as D
        n, rp_t, ci_t, va_t = (D.orkut_like if kind == "orkut" else D.wikitalk_like)(device="cuda" if torch.cuda.is_available() else "cpu")
        rp, ci, va = rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy()
        return n, n, rp, ci, va, ("synthetic com-Orkut-shaped (symmetric, mean degree ~70), seed 20261004" if kind == "orkut"
                                  else "synthetic wiki-Talk-shaped (94 % empty rows, rows up to 100 000), seed 20261005")
    if kind != "webgoogle":
        sys.exit(f"unknown workload {kind!r}: webgoogle | livejournal | orkut | wikitalk | rmat<scale> | banded<rows>")
    f = synth.data_file("web-Google.mtx")
    if f:
        # the reference's own CSR of this file, bit for bit (REFCOMPAT loader: 1-based arrays taken literally, values idx % 13 in
        # FILE order for a pattern file, spmv.cpp:413-417; the pad-to-16 copies carry the value 0) -- what spmv.cvr runs
        m = cvr_amd.load_mm(f, capi.MM_REFCOMPAT)
        return m["nrows"], m["ncols"], m["row_ptr"], m["col_idx"], m["vals"], "web-Google.mtx (SNAP)"
    n, nc, rp, ci, va = synth.web_google_like()
    return n, nc, rp, ci, va, "synthetic web-Google-shaped, seed 20261002"


def _child_env(**extra):
    """the environment of the CPU baseline's child processes: this process's, without what a profiler preloaded into it (LD_PRELOAD,
    ROCP_* / ROCPROFILER_* / HSA_TOOLS_*): a child that inherits those is a second profiled process, and a launcher that execs its
    target (numactl) would be an exec of a process the profiler has already initialised the GPU in"""
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF", "HSA_TOOLS_", "ROCTRACER_"))}
    env.update(extra)
    return env


def _profiled():
    """under rocprofv3 (its tool library is preloaded; the GPU boxes preload an exec guard of their own into every process, which is not that)"""
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCP_", "ROCPROF", "HSA_TOOLS_")) for k in os.environ)


def _numactl_prefix():
    """the reference's own recipe binds memory (run_sample.sh:10: numactl --membind); here, where the threads span the sockets,
    the pages are interleaved over all nodes so that a run does not depend on which socket first-touched them.  Not under a profiler
    (no launcher hop there: first-touch placement; profile with --no-cpu-baseline anyway)"""
    import shutil
    if _profiled():
        return [], "run under a profiler: no numactl launcher, first-touch placement"
    exe = shutil.which("numactl")
    if not exe:
        return [], "no numactl on this host: first-touch placement"
    try:
        ok = subprocess.run([exe, "--interleave=all", "true"], capture_output=True, timeout=20, env=_child_env()).returncode == 0
    except Exception:
        ok = False
    return ([exe, "--interleave=all"], "numactl --interleave=all") if ok else ([], "numactl refused --interleave=all: first-touch placement")


def _cpu_topology():
    """{package id: [one logical CPU per physical core, ascending]} of the CPUs this process may run on (sysfs); {} if it cannot be read"""
    try:
        allowed = os.sched_getaffinity(0)
    except Exception:
        return {}
    pk = {}
    for c in sorted(allowed):
        base = f"/sys/devices/system/cpu/cpu{c}/topology/"
        try:
            pkg = int(open(base + "physical_package_id").read())
            sib = open(base + "thread_siblings_list").read().strip()
            first = int(sib.replace("-", ",").split(",")[0])
        except Exception:
            return {}
        if first == c or first not in allowed:          # the first hardware thread of its core (or the only one this process may use)
            pk.setdefault(pkg, []).append(c)
    return pk


def _cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us); None: no limit.  Round 5: the
    GPU boxes of this pool show 256 logical CPUs and a quota of 16 -- 64 reference threads were throttled in 80 % of the scheduler periods
    (cpu.stat nr_throttled), which is where a spread of a factor of three between identical runs came from."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(int(q) / int(per)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, q // per)
    except Exception:
        return None


def _run_reference(exe, path, T, iters, nnz, nrows, ncols, prefix=(), cpus=None):
    """one run of the reference binary; cpus: the CPUs the child is pinned to before it starts (its loader thread first-touches every
    page there, its OpenMP threads are placed inside the set: what `numactl --membind` does for the reference, run_sample.sh:10)"""
    import re
    from cvr_amd import synth
    env = _child_env(OMP_NUM_THREADS=str(T), OMP_PROC_BIND="close", OMP_PLACES="cores")
    pin = (lambda: os.sched_setaffinity(0, cpus)) if cpus else None
    try:
        r = subprocess.run(list(prefix) + [exe, path, str(T), str(iters)], capture_output=True, text=True, timeout=240, env=env, preexec_fn=pin)
    except Exception:
        return None
    out = r.stdout
    m = re.search(r"SpMV Execution Time of CVR\s+is ([0-9.eE+-]+) seconds", out)
    p = re.search(r"Pre-processing\(CSR->CVR\)\s+Time of CVR\s+is ([0-9.eE+-]+) seconds", out)
    if r.returncode != 0 or not m or "Very Good" not in out:
        return None
    per = float(m.group(1))
    return {"threads": T, "ms_per_step": per * 1e3, "gflops": 2.0 * nnz / per / 1e9, "reference_convention_gflops": nnz / per / 1e9,
            "gbs_alg": synth.b_alg(nrows, ncols, nnz) / per / 1e9, "preprocess_s": float(p.group(1)) if p else None, "pinned_cpus": len(cpus) if cpus else 0}


def _cpu_reference(nrows, ncols, rp, ci, logical, physical):
    """the UNMODIFIED reference (spmv.cpp compiled by oracle/Makefile into oracle/_ref/, prebuilt in the build container) on a
    Matrix-Market file of the bench matrix: kind = "reference".  None if it cannot run here.  Two configurations, five runs each:
    (a) ONE SOCKET: min(68, physical cores of a socket) threads, the process pinned to that socket's cores (one hardware thread per
    core) before it starts -- threads and pages on one socket, as run_sample.sh:10 binds its 68 threads' memory to one node --, and
    (b) one thread per physical core of the whole host, pinned to exactly those."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    exe = os.path.join(ROOT, "oracle", "_ref", "spmv.cvr.ref")
    if not os.path.exists(exe):
        return None
    iters = 1000         # (a second of SpMVs per run: round 5's first pinned runs of 300 iterations differed by a factor of three on a host shared with other jobs)
    repeats = 5
    def _host_state():          # what else runs on the host while the baseline is timed (the GPU boxes of this pool are containers on a shared host)
        st = {}
        try:
            st["loadavg"] = open("/proc/loadavg").read().split()[:3]
        except OSError:
            pass
        for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat"):
            try:
                st[os.path.basename(f)] = " ".join(open(f).read().split())
            except OSError:
                pass
        return st
    host_before = _host_state()
    runs = []
    topo = _cpu_topology()
    configs = []          # (label, threads, cpus or None)
    quota = _cpu_quota()
    if topo:
        pkg0 = sorted(topo)[0]
        room = max(1, quota - 1) if quota else 68          # (never more threads than the container's CPU quota -- they would be throttled, not run --, and one CPU left to the parent and the children's loaders: with all 16 of 16 taken, 15 scheduler periods of a run were still throttled)
        one = topo[pkg0][:min(68, room)]
        every = sorted(c for cs in topo.values() for c in cs)
        configs.append((f"one socket: {len(one)} threads on {len(one)} cores of package {pkg0}" + (f" (the container's CPU quota: {quota})" if quota and quota < 68 else ""), len(one), set(one)))
        if len(every) > len(one) and (quota is None or quota >= len(every)):
            configs.append((f"all {len(every)} physical cores of {len(topo)} packages", len(every), set(every)))
        elif quota and len(topo) > 1 and quota < len(every):          # the same number of threads over all sockets (more memory channels, remote pages)
            spread_set = sorted(c for cs in topo.values() for c in cs[:max(1, room // len(topo))])
            configs.append((f"{len(spread_set)} threads over {len(topo)} packages", len(spread_set), set(spread_set)))
        placement = "child pinned with sched_setaffinity before it starts (threads and first-touched pages inside the set)"
        prefix = ()
    else:
        prefix, placement = _numactl_prefix()
        for T in sorted({min(logical, 68), physical}, reverse=True):
            configs.append((f"{T} threads, unpinned", T, None))
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        path = os.path.join(d, "bench.mtx")
        O.write_mtx_pattern(path, nrows, ncols, rp, ci)
        for label, T, cpus in configs:
            for _ in range(repeats):
                r = _run_reference(exe, path, T, iters, len(ci), nrows, ncols, prefix, cpus)
                if r:
                    r["config"] = label
                    runs.append(r)
    if not runs:
        return None
    # per configuration: min / median / max over the repeats and their spread; `value` is the MEDIAN run of the configuration whose
    # median is better -- a typical run, not the luckiest
    by_c = {}
    for r in runs:
        by_c.setdefault(r["config"], []).append(r)
    stats = []
    for label, rs in by_c.items():
        ms = sorted(x["ms_per_step"] for x in rs)
        med = ms[len(ms) // 2]
        stats.append({"config": label, "threads": rs[0]["threads"], "runs": len(rs), "ms_per_step_min": ms[0], "ms_per_step_median": med, "ms_per_step_max": ms[-1],
                      "spread_max_minus_min_over_median": (ms[-1] - ms[0]) / med})
    pick = min(stats, key=lambda x: x["ms_per_step_median"])
    chosen = sorted(by_c[pick["config"]], key=lambda x: x["ms_per_step"])
    best = chosen[len(chosen) // 2]
    return {"value": best["gflops"], "unit": "GFLOP/s", "cores": best["threads"], "kind": "reference",
            # the reference's own configuration is 68 threads on one node (run_sample.sh:10); a container with a CPU quota runs as many threads as the quota leaves room for
            "baseline_threads": best["threads"], "reference_threads": 68, "quota_limited": bool(quota and quota < 68),
            "sample": f"unmodified reference source built by oracle/Makefile (g++ -O3 -mavx512f -fopenmp, 4 intrinsic-spelling aliases in oracle/ref_shim.h), "
                      f"{iters} timed SpMV iterations of the full matrix per run, y zeroing outside the timer as the reference does (spmv.cpp:1026-1033); "
                      f"{repeats} runs per configuration ({'; '.join(c[0] for c in configs)}), memory: {placement}; "
                      "`value` is the median run of the configuration whose median is better",
            "ms_per_step": best["ms_per_step"], "preprocess_s": best["preprocess_s"], "gbs_alg": best["gbs_alg"], "spread": pick["spread_max_minus_min_over_median"],
            "reference_convention_gflops": best["reference_convention_gflops"], "per_thread_count": stats, "memory_placement": placement, "runs": runs,
            "host_state": {"before": host_before, "after": _host_state(), "logical_cpus_visible": len(os.sched_getaffinity(0)), "cpu_quota": quota}}


def _cpu_port(nrows, ncols, rp, ci, va, cores, budget_s=8.0, probe=(8, 16, 32, 64, 128)):
    """the oracle's scalar-C restatement of the reference CPU path (8 lanes, one chunk per OpenMP thread; spmv.cpp:565-1014,
    1016-1667) on the reference loader's 1-based arrays, y zeroing INSIDE the timer: kind = "port" """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    from cvr_amd import synth
    m = synth.to_refcompat(nrows, ncols, rp, ci, va)
    x = np.ones(ncols + 2)
    best = None
    quota = _cpu_quota()
    if quota:          # (threads beyond the container's CPU quota are throttled, not run)
        cores = min(cores, quota)
    for T in sorted({max(1, min(cores, t)) for t in probe}):     # thread count: quick probe, keep the best
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
    """SURVEY 8(d) "CPU baseline beside it": the reference on the host cores (zeroing outside its timer, as it does), and a
    second figure with the zeroing inside the timer (the oracle's port of the same path: the unmodified reference cannot move it)"""
    model, logical, physical = host_cpu()
    r = None
    if os.environ.get("CVR_CPU_BASELINE", "reference") == "reference":
        r = _cpu_reference(nrows, ncols, rp, ci, logical, physical)
    if r is None:
        r = _cpu_port(nrows, ncols, rp, ci, va, logical)
    elif not os.environ.get("CVR_BENCH_NO_SECOND_CPU_FIGURE"):
        try:
            second = _cpu_port(nrows, ncols, rp, ci, va, logical, budget_s=4.0, probe=(min(logical, 68), physical))
            if second:
                r["zeroing_inside_timer"] = {k: second[k] for k in ("value", "unit", "cores", "kind", "ms_per_step", "sample")}
        except Exception as e:
            r["zeroing_inside_timer"] = {"error": repr(e)}
    if r:
        r["host"] = {"cpu_model": model, "logical_cores": logical, "physical_cores": physical}
    return r


def _visible_gpus():
    """GPUs this process could open, from the KFD topology in sysfs (nodes with SIMDs) and the *_VISIBLE_DEVICES masks -- no HIP
    or HSA call, so the parent of the ranks never initialises the GPU.  None when sysfs does not tell (the ranks then report)."""
    import glob
    n = 0
    try:
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(f):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except Exception:
        return None
    if n == 0:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def self_launch(args):
    """N > 1 without a torch.distributed environment: one child per GPU via torch.distributed.run, started before this
    process has touched the GPU (the box refuses an exec / fork from a process that has); rank 0's JSON line is relayed."""
    have = _visible_gpus()                    # counted without a HIP call: this process must stay off the GPU
    if have is not None and have < args.gpus and not os.environ.get("CVR_BENCH_ONE_DEVICE"):
        sys.exit(f"bench.py --gpus {args.gpus}: {have} GPU(s) visible (CVR_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 as a plumbing check)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // args.gpus)))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line)
    sys.exit(r.returncode if r.returncode else (0 if line else 1))


def _pre_times(info):
    """the preprocessing times of a handle (cvr_info): T_pre = analysis + conversion without the upload, as tools/compare_csr.py counts it"""
    return {"plan_s": info.plan_s, "probe_s": info.probe_s, "upload_s": info.upload_s, "convert_s": info.convert_s,
            "preprocess_wall_s": info.preprocess_wall_s, "dict_s": info.dict_s, "one_submission": bool(info.preprocess_fused),
            "t_pre_s": info.plan_s + info.probe_s + info.hub_select_s + info.dict_s + info.preprocess_wall_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--steps-per-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--col-panels", type=int, default=-1, help="column panels (-1 = the library's rule)")
    ap.add_argument("--interleave", type=int, default=-1, help="interleaved chunks (cvr_options.interleave; -1 = the library's rule)")
    ap.add_argument("--partition", default="cost", choices=["cost", "nnz"], help="row shards balanced by predicted time (non-zeros + 1.25 per row: cvr_row_partition_cost) "
                    "or by non-zeros alone (the reference's rule)")
    ap.add_argument("--other-workloads", default="auto", help="comma-separated shapes measured beside the headline in the same JSON line (N = 1, headline workload only); "
                    "auto = " + ",".join(OTHER_WORKLOADS) + " within a time box; none = skip")
    ap.add_argument("--two-streams", action="store_true", help="also report the throughput of independent SpMVs alternating on two streams "
                    "(off by default: concurrent kernels would distort a rocprofv3 kernel-time summary of this command)")
    ap.add_argument("--workload", default="webgoogle", help="webgoogle (the headline, default) | livejournal | orkut | wikitalk | banded[<rows>] | rmat[<scale>] (fp32)")
    ap.add_argument("--dump-y", default="", help="rank 0 writes the y of the last timed step (N > 1: the all-gathered vector, padding removed) to this .npy file: "
                    "the tests check it against the oracle (bench.py itself may not use oracle/ outside its cpu_baseline leg)")
    ap.add_argument("--emulate-rank", default="", help="r/N with --gpus 1 and a device-built workload (rmat<scale>, banded<rows>): build and time ONLY rank r's "
                    "row shard of an N-way partition (x replicated, as on N GPUs); no exchange.  The per-rank regime of the 8-GPU configurations on one GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)

    import torch
    import cvr_amd
    from cvr_amd import shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # "nccl" is RCCL on ROCm; RCCL refuses two ranks on one GPU, so the one-device plumbing check runs over gloo
    backend = os.environ.get("CVR_BENCH_BACKEND", "gloo" if os.environ.get("CVR_BENCH_ONE_DEVICE") else "nccl")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU fallback")
    if os.environ.get("CVR_BENCH_ONE_DEVICE"):      # plumbing check of the N > 1 path on a 1-GPU box (all ranks on cuda:0)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    sharded = world > 1 or bool(os.environ.get("CVR_BENCH_FORCE_SHARDED"))   # the latter: the N > 1 code path with one rank (RCCL plumbing check)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- the workload: host arrays (web-Google, LiveJournal shapes) or shard-local device arrays (banded, R-MAT) ----
    t_build0 = time.perf_counter()
    device_built = args.workload.startswith("rmat") or args.workload.startswith("banded")
    tune = world > 1 and args.steps_per_chunk == 0 and not os.environ.get("CVR_BENCH_NO_TUNE")
    rp = ci = va = None
    row_cost = 0 if args.partition == "nnz" else capi_row_cost()
    pworld, prank = world, rank           # the partition the shard belongs to (--emulate-rank: one rank of N on this one GPU)
    if args.emulate_rank:
        if world != 1 or not device_built:
            sys.exit("--emulate-rank needs --gpus 1 and a device-built workload (rmat<scale> / banded<rows>)")
        prank, pworld = (int(v) for v in args.emulate_rank.split("/"))
        if not 0 <= prank < pworld:
            sys.exit("--emulate-rank r/N: 0 <= r < N")
    if device_built:
        from cvr_amd import synth_dev as D
        if args.workload.startswith("rmat"):
            scale = int(args.workload[4:] or 22)
            nrows = ncols = 1 << scale
            deg = D.rmat_row_degrees(scale, device=dev)
            bounds, grp = D.partition_from_degrees(deg, pworld, row_cost)
            nnz = int(grp[-1])
            nnz_per = [int(grp[bounds[p + 1]] - grp[bounds[p]]) for p in range(pworld)]
            del deg, grp
            lrp_t, lci_t, lva_t = D.rmat_rows(scale, int(bounds[prank]), int(bounds[prank + 1]), device=dev)
            source = f"synthetic R-MAT scale {scale}, edge factor 16, fp32, built shard by shard on the GPU (torch generator streams: cvr_amd/synth_dev.py)"
            f32 = True
        else:
            nrows = ncols = int(float(args.workload[6:] or 3.5e6))
            bounds, nnz = D.banded_partition(nrows, 13, pworld, row_cost)
            lrp_t, lci_t, lva_t = D.banded_rows(nrows, int(bounds[prank]), int(bounds[prank + 1]), device=dev)
            r = np.arange(nrows, dtype=np.int64)
            pre = np.concatenate([[0], np.cumsum(np.minimum(r, 13) + 1 + np.minimum(nrows - 1 - r, 13))])
            nnz_per = [int(pre[bounds[p + 1]] - pre[bounds[p]]) for p in range(pworld)]
            del r, pre
            source = "synthetic banded symmetric (27 nnz/row, nlpkkt240's shape), built shard by shard on the GPU (cvr_amd/synth_dev.py)"
            f32 = False
        torch.cuda.synchronize()
        lrows, lnnz = int(bounds[prank + 1] - bounds[prank]), int(lrp_t[-1])
        build_s = time.perf_counter() - t_build0
        A = cvr_amd.CvrMatrix.from_device(lrows, ncols, lrp_t.data_ptr(), lci_t.data_ptr(), lva_t.data_ptr(), is_f32=f32, device=local_rank,
                                          steps_per_chunk=args.steps_per_chunk, tune_steps=False, col_panels=args.col_panels, interleave=args.interleave)      # (device-built large workloads keep the library's rules: hub tables, panels; tuning is for shards of the small headline matrix)
        np_dtype = np.float32 if f32 else np.float64
    else:
        nrows, ncols, rp, ci, va, source = load_host_workload(args.workload)
        nnz = len(ci)
        bounds = shard.row_partition(rp, world, row_cost)
        nnz_per = [int(rp[bounds[p + 1]] - rp[bounds[p]]) for p in range(world)]
        lrows, lrp, lci, lva = shard.local_csr(rp, ci, va, bounds, rank)
        lnnz = int(lrp[-1])
        build_s = time.perf_counter() - t_build0
        # a shard of this matrix on one of N GPUs is small enough for the layout to matter: measure it (cvr_tune)
        A = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, device=local_rank, steps_per_chunk=args.steps_per_chunk, tune_steps=tune, col_panels=args.col_panels, interleave=args.interleave)
        f32 = va.dtype == np.float32
        np_dtype = va.dtype
    create_s = time.perf_counter() - t_build0 - build_s
    info = A.info
    # The first cvr_create of a process also loads the code object and creates the library's streams (milliseconds); the preprocessing
    # times DESIGN section 5.7 / 5.11 and tools/compare_csr.py quote are those of a process that has done so.  Host workloads that build in
    # a moment are therefore built a second time, untimed by the bench, and the JSON line carries both.
    warm_info = None
    if not (device_built or tune) and create_s < 2.0:
        t_w0 = time.perf_counter()
        A2 = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, device=local_rank, steps_per_chunk=args.steps_per_chunk, tune_steps=False, col_panels=args.col_panels, interleave=args.interleave)
        warm_create_s = time.perf_counter() - t_w0
        warm_info = A2.info
        A2.close()
    max_rows, pick = shard.gather_layout(bounds)

    tdt, vbytes = (torch.float32, 4) if f32 else (torch.float64, 8)
    bits = torch.int32 if f32 else torch.int64
    x = torch.zeros(info.x_elems, dtype=tdt, device=dev)
    x[:ncols] = torch.from_numpy(synth.x_rand(ncols, np_dtype)).to(dev)
    xcheck = lambda tag: None
    if os.environ.get("CVR_BENCH_DEBUG_DUMP"):          # (diagnostics: is x still what was uploaded?)
        x_host_dbg = synth.x_rand(ncols, np_dtype)
        def xcheck(tag):
            if tag in os.environ.get("CVR_BENCH_DEBUG_SKIP", ""):
                return
            torch.cuda.synchronize()
            got = x[:ncols].cpu().numpy()
            d_ = np.flatnonzero(got != x_host_dbg)
            if len(d_):
                print(f"[bench rank {rank}] x differs at \"{tag}\": {len(d_)} of {ncols} elements, first {d_[0]} last {d_[-1]}, got {got[d_[:4]].tolist()} want {x_host_dbg[d_[:4]].tolist()}, "
                      f"zeros among them {int((got[d_] == 0).sum())}, x ptr {x.data_ptr():#x}", file=sys.stderr, flush=True)
            else:
                print(f"[bench rank {rank}] x ok at \"{tag}\"", file=sys.stderr, flush=True)
    xcheck("uploaded")
    ny = max(info.yext_elems, max_rows)
    ybufs = [torch.zeros(ny, dtype=tdt, device=dev) for _ in range(2 if sharded else 1)]
    yalls = [torch.zeros(world * max_rows, dtype=tdt, device=dev) for _ in range(2)] if sharded else None
    y = ybufs[0]
    # Everything above (the upload of x, the zero fills) was enqueued on the device's default stream, and the stream below does not wait for
    # that stream: drain it before the first launch.  (Round 5, eight processes on one device: the first SpMV of the slowest rank ran on the
    # new stream BEFORE the default stream had copied x out of its staging tensor, whose memory had by then become a y buffer -- x came
    # out as that buffer's zeros from then on, and the run's own check, made with the same x, passed: profiles/r05_eight_ranks_debug.log.)
    torch.cuda.synchronize()
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
        if os.environ.get("CVR_BENCH_GATHER", "native") == "native" and backend == "nccl":
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

    xcheck("buffers, stream")
    rccl_info = None
    if comm is not None:
        try:
            rn, rr, rv = comm.info()
            rccl_info = {"ranks": rn, "rank0_user_rank": rr, "version": rv}
        except Exception as e:          # noqa: BLE001
            rccl_info = {"error": repr(e)}
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
            torch.cuda.synchronize()          # (one rank alone: nothing new to drain -- a synchronize of an idle device takes ~10 us, 2 % of a 20-step run of the headline)

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
    xcheck("warmup")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); e1.record(stream)          # (a torch event is created at its first record: not inside the timed region)
    sync()
    t0 = time.perf_counter()
    e0.record(stream)
    t_a = time.perf_counter()
    step(args.steps)
    t_b = time.perf_counter()
    e1.record(stream)
    while not e1.query():        # (poll: a blocking wait wakes this thread 10-20 us after the last kernel has ended, which is 5 % of a 20-step run of the headline)
        pass
    t_c = time.perf_counter()
    sync()
    wall = time.perf_counter() - t0
    timed_region_us = {"event_record": (t_a - t0) * 1e6, "launch_calls_returned": (t_b - t0) * 1e6, "last_kernel_done_seen": (t_c - t0) * 1e6, "after_synchronize": wall * 1e6}
    ev_s = e0.elapsed_time(e1) * 1e-3
    if sharded:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # the SpMV kernel alone, HIP events on the launch stream (N > 1: this rank's shard, no gather)
    def kernel_time(M, ybuf):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # behind ~30 ms of untimed launches: after an idle spell the chip needs some ten milliseconds of work before a kernel takes its steady time (the headline
        # behind a 20-step run: 20.93 us over the next 200 launches, 20.6 behind 1 000 steps -- what rocprofv3 averages over a whole run; profiles/r06_timed_region_20_steps.log)
        n_w = max(args.warmup, min(2000, int(0.03 / max(wall / max(args.steps, 1), 1e-6))))
        M.spmv_device(x.data_ptr(), ybuf.data_ptr(), sptr, repeat=n_w)
        n_t = max(args.steps, 200)               # >= 100 timed back-to-back launches, whatever --steps says
        a.record(stream)
        M.spmv_device(x.data_ptr(), ybuf.data_ptr(), sptr, repeat=n_t)
        b.record(stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / n_t
    kern_s = kernel_time(A, y)
    xcheck("kernel_time")
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
    # of the other); reported for information, never the value: the steps of the metric run one after the other.  A handle with
    # column panels keeps its partial sums in one buffer: one SpMV in flight at a time (include/cvr_amd.h), so no such figure.
    two = None
    if not sharded and args.two_streams and info.col_panels == 1:
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
    balg_local = synth.b_alg(lrows, ncols, lnnz, vbytes)
    achieved = balg_local / kern_s / 1e9
    streamed_local = int(info.image_bytes) + (ncols + lrows) * vbytes      # what this layout moves: image + x + y

    # the same matrix with full values in the stream (value_dict = 0): separates the format from the compression of the 13
    # distinct values the reference's loader gives a pattern file (spmv.cpp:417)
    kern_nodict_s = None
    if not sharded and not device_built and info.value_dict > 0 and not os.environ.get("CVR_BENCH_NO_DICT_OFF_RUN"):
        try:
            B = cvr_amd.CvrMatrix(lrows, ncols, lrp, lci, lva, device=local_rank, steps_per_chunk=args.steps_per_chunk, value_dict=0, col_panels=args.col_panels, interleave=args.interleave)
            kern_nodict_s = kernel_time(B, torch.zeros(max(B.info.yext_elems, 1), dtype=tdt, device=dev))
            B.close()
        except Exception as e:
            print(f"[bench] value_dict = 0 run failed: {e!r}", file=sys.stderr)

    # parity guard on the timed configuration: y of the last step against the CSR loop (the reference's own self-check,
    # spmv.cpp:1843-1850, 1916-1938): the product's host loop for host-built workloads, a torch fp64 loop on the rank's own
    # shard for device-built ones
    wrong = wrong_ref = -1
    x_damaged = int(torch.count_nonzero(x[:ncols] != torch.from_numpy(synth.x_rand(ncols, np_dtype)).to(dev)).item())     # (the device reference below is made from this x)
    if x_damaged:
        print(f"[bench rank {rank}] x on the device differs from what was uploaded in {x_damaged} elements", file=sys.stderr)
    if device_built:
        from cvr_amd import synth_dev as D
        yref_t, absy_t = D.csr_spmv_reference(lrp_t, lci_t, lva_t, x[:ncols])
        tol_t = 1e-5 if f32 else 1e-12
        if sharded:
            # every rank checks the WHOLE gathered vector: the owners' references travel the same way as y (padded slices, all-gathered), so a
            # rank's copy of another rank's slice is compared too -- the exchange is part of what is verified, not only the local SpMV
            pad = torch.zeros(2 * max_rows, dtype=torch.float64, device=dev)
            pad[:lrows] = yref_t
            pad[max_rows: max_rows + lrows] = absy_t
            allref = torch.zeros(world * 2 * max_rows, dtype=torch.float64, device=dev)
            shard.all_gather_tensor(allref, pad)
            torch.cuda.synchronize()
            allref = allref.view(world, 2, max_rows)
            yg = yalls[last[0]].view(world, max_rows).to(torch.float64)
            bad = torch.zeros(1, dtype=torch.int64, device=dev)
            for p in range(world):
                np_ = int(bounds[p + 1] - bounds[p])
                bad += torch.count_nonzero((yg[p, :np_] - allref[p, 0, :np_]).abs() > tol_t * allref[p, 1, :np_] + 1e-300)
            bad += x_damaged                                       # (a rank whose x is not the uploaded one fails the run whatever its y)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)          # (the worst rank's count of wrong rows over the whole vector)
        else:
            yl = y[:lrows].to(torch.float64)
            bad = torch.count_nonzero((yl - yref_t).abs() > tol_t * absy_t + 1e-300).to(torch.int64).reshape(1) + x_damaged
        wrong = int(bad.item())
    else:
        yh = (yalls[last[0]][torch.from_numpy(pick).to(dev)] if sharded else y[:nrows]).cpu().numpy()
        if rank == 0:
            xh = synth.x_rand(ncols, np_dtype).astype(np.float64)          # (what was uploaded, not what the device holds now)
            nt = len(os.sched_getaffinity(0))
            yref = cvr_amd.csr_spmv_host(rp, ci, va.astype(np.float64), xh, nthreads=nt)
            # rows off by more than tol * sum |a x| (1e-12 fp64, 1e-5 for the fp32 path, which has no reference
            # counterpart), and beside it the reference's own criterion (abs 1e-3, spmv.cpp:1916-1938)
            absy = cvr_amd.csr_spmv_host(rp, ci, np.abs(va).astype(np.float64), np.abs(xh), nthreads=nt)
            tol = 1e-5 if f32 else 1e-12
            wrong = int(np.count_nonzero(np.abs(yh.astype(np.float64) - yref) > tol * absy + (1e-30 if f32 else 1e-300)))
            wrong_ref = int(cvr_amd.verdict(yh.astype(np.float64), yref, nrows))

    # the exchange itself: every rank must hold the SAME gathered vector, and rank p's copy of slice p is the one its own check above
    # covered -- so per slice, every rank's bits against the owner's (a 64-bit wrap-around sum of the slice's bit patterns)
    gathered_mismatch = None
    shard_sums = None
    if sharded and device_built:          # what every rank built (the tests compare it with slices of the matrix built in one piece)
        mine_s = [int(lrows), int(lnnz), int(lci_t.to(torch.int64).sum().item()), int((lva_t.to(torch.float64) * 1e6).round().to(torch.int64).sum().item())]
        shard_sums = [None] * world
        dist.all_gather_object(shard_sums, mine_s)
    if sharded:
        sizes = [int(bounds[p + 1] - bounds[p]) for p in range(world)]
        mine = torch.stack([yalls[last[0]][p * max_rows: p * max_rows + sizes[p]].view(bits).to(torch.int64).sum() for p in range(world)])
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        tab = torch.stack([e.cpu() for e in every]).numpy()          # [rank][slice]
        gathered_mismatch = int(sum(int(tab[r][p] != tab[p][p]) for r in range(world) for p in range(world)))
        if gathered_mismatch and rank == 0:
            print("[bench] gathered y differs between ranks: (rank, slice) pairs off: "
                  + ", ".join(f"({r},{p})" for r in range(world) for p in range(world) if tab[r][p] != tab[p][p]), file=sys.stderr)
    xcheck("verified")
    if args.dump_y and os.environ.get("CVR_BENCH_DEBUG_DUMP") and sharded and device_built:
        # (diagnostics, every rank: its shard's arrays hashed, its own y buffers, its reference, its copy of the gathered vector)
        import hashlib
        torch.cuda.synchronize()
        hs = [hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest() for t in (lrp_t, lci_t, lva_t)]
        np.savez(args.dump_y + f".rank{rank}.npz", hashes=np.asarray(hs), y0=ybufs[0][:lrows].cpu().numpy(), y1=ybufs[1][:lrows].cpu().numpy(),
                 yref=yref_t.cpu().numpy(), absy=absy_t.cpu().numpy(), yall0=yalls[0].cpu().numpy(), yall1=yalls[1].cpu().numpy(), last=np.asarray([last[0]]),
                 x=x[:ncols].cpu().numpy())
    if args.dump_y and rank == 0:
        torch.cuda.synchronize()
        ydump = (yalls[last[0]][torch.from_numpy(pick).to(dev)] if sharded else y[:nrows]).cpu().numpy()
        np.save(args.dump_y, ydump)
        if os.environ.get("CVR_BENCH_DEBUG_DUMP") and sharded:          # (diagnostics: the padded gathered buffer, the index array and the bounds as they are)
            np.save(args.dump_y + ".raw.npy", yalls[last[0]].cpu().numpy())
            np.save(args.dump_y + ".pick.npy", np.asarray(pick))
            np.save(args.dump_y + ".bounds.npy", np.asarray(bounds))
    copy_gbs = None
    if rank == 0:
        try:                      # achievable-HBM yardstick measured live: 1 GiB streaming copy, read + write
            from cvr_amd import capi
            copy_gbs = capi.device_copy_gbs(local_rank, 1 << 30, 20)
        except Exception:
            copy_gbs = None
    if rank == 0:
        per = wall / args.steps
        emu = bool(args.emulate_rank)
        job_nnz = lnnz if emu else nnz            # (an emulated rank: the flops of its shard)
        kname = ("cvr::spmv_gang_kernel" if info.gang else "cvr::spmv_ilv_kernel" if info.interleave else "cvr::spmv_seg_kernel" if info.col_phases > 1 else "cvr::spmv_kernel") + ("<float>" if f32 else "<double>")      # (column phases: the kernel without hand-out state; interleaved chunks: the hand-pipelined one)
        workload_text = f"{source}: {nrows}x{ncols}, nnz {nnz}, {'fp32' if f32 else 'fp64'}, y = A x with A (CVR64 image), x, y resident in HBM"
        out = {
            "metric": "SpMV GFLOP/s (2*nnz/t), web-Google fp64" if args.workload == "webgoogle" else f"SpMV GFLOP/s (2*nnz/t), {args.workload} {'fp32' if f32 else 'fp64'}"
                      + (f", rank {prank} of {pworld} alone on one GPU (its row shard, x replicated, no exchange)" if emu else ""),
            "value": 2.0 * job_nnz / per / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if f32 else "f64",
            "data": "synthetic" if source.startswith("synthetic") else "real",
            "config": {"workload": workload_text,
                       "rows_per_gpu": [int(v) for v in np.diff(bounds)],
                       "nnz_per_gpu": nnz_per,
                       "nnz_imbalance_max_over_mean": float(max(nnz_per) * pworld / max(nnz, 1)),
                       "emulated_rank": [prank, pworld] if emu else None, "rank_rows": lrows, "rank_nnz": lnnz,
                       "steps_per_chunk": int(info.steps_per_chunk), "chunks_rank0": int(info.nchunks),
                       "rows_cut_rank0": int(info.nshared), "col_panels": int(info.col_panels),
                       "waves_per_workgroup": int(info.waves_per_block), "x_window_values": int(info.x_window), "col_phases": int(info.col_phases),
                       "lds_bytes_per_workgroup": int(info.lds_bytes), "near_diagonal_share": float(info.near_diagonal_share),
                       "value_dictionary_entries": int(info.value_dict),
                       "parallelism": ("rows sharded, x replicated, y all-gathered (%s)" % ("RCCL" if backend == "nccl" else backend)) if sharded else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, **(pmc_fields(info, kname, workload_text, lnnz) if world == 1 else {"traffic": None}),
                         "achieved_is": "algorithmic bytes of SURVEY 8(d) (12 B per non-zero for fp64, whatever the image stores) / kernel time",
                         "kernel": kname, "kernel_us": kern_s * 1e6,
                         "kernel_us_median_single_launches": singles[len(singles) // 2] if singles else None, "kernel_us_min_single_launches": singles[0] if singles else None,
                         "copy_kernel_gbs": copy_gbs, "frac_of_copy_kernel": achieved / copy_gbs if copy_gbs else None,
                         # the boxes of the pool run stream-bound kernels in one of two modes ~17 % apart (profiles/r05_bimodal_probe.log); the copy kernel
                         # lands in the same mode as the process's other streaming kernels: which one this line was taken in
                         "box_mode": None if not copy_gbs else ("fast (copy kernel >= 5.6 TB/s)" if copy_gbs >= 5600 else "slow (copy kernel < 5.6 TB/s)"),
                         "algorithmic_bytes_per_launch": int(balg_local),
                         "streamed_bytes_per_launch": streamed_local, "streamed_gbs": streamed_local / kern_s / 1e9,
                         "kernel_us_value_dict_off": None if kern_nodict_s is None else kern_nodict_s * 1e6,
                         "frac_value_dict_off": None if kern_nodict_s is None else balg_local / kern_nodict_s / 1e9 / HBM_PEAK_GBS},
            "image_bytes": int(info.image_bytes),
            "gbs_alg_whole_job": synth.b_alg(lrows if emu else nrows, ncols, job_nnz, vbytes) / per / 1e9,
            "event_ms_per_step_rank0": ev_s / args.steps * 1e3,
            "timed_region_host_clock_us": timed_region_us,          # where the K steps' wall time goes beside K kernels: host time stamps from the start of the timed region
            "spmv_only_ms_max_over_ranks": kern_max_s * 1e3, "gflops_spmv_only_no_exchange": 2.0 * job_nnz / kern_max_s / 1e9,
            "rank0_spmv_only_ms": kern_s * 1e3, "rank0_allgather_only_ms": None if gather_s is None else gather_s * 1e3,
            "preprocess": {**_pre_times(info), "tune_s": A.tuning_s, "workload_build_s": build_s, "create_and_preprocess_wall_s": create_s,
                           "which": "the first cvr_create of the process (code object load, stream creation included)",
                           "warm": None if warm_info is None else {**_pre_times(warm_info), "create_and_preprocess_wall_s": warm_create_s,
                                                                    "which": "the same matrix built a second time in this process"}},
            "independent_spmvs_on_two_streams": None if two is None else {"ms_per_spmv": two * 1e3, "gflops": 2.0 * nnz / two / 1e9},
            "gathered_slices_differing_between_ranks": gathered_mismatch, "shard_checksums": shard_sums,
            "verdict_wrong_rows": wrong, "verdict_tolerance": "rows with |y - y_csr| > %g * sum |a x|" % (1e-5 if f32 else 1e-12),
            "verdict_wrong_rows_reference_criterion_abs_1e-3": wrong_ref if wrong_ref >= 0 else None, "gather_impl": gather_impl, "gather_calibration_ms_per_step": calib,
            # what RCCL itself reports for the library's communicator (ncclCommCount / ncclCommUserRank / ncclGetVersion): a multi-GPU record shows that the collective saw N ranks
            "rccl": rccl_info,
        }
        # the other power-law shapes, in this process, time-boxed (the default run must finish within minutes): each entry lets the
        # fractions be recomputed from nnz and kernel time
        if world == 1 and args.workload == "webgoogle" and not args.emulate_rank and args.other_workloads != "none":
            names = OTHER_WORKLOADS if args.other_workloads == "auto" else tuple(v for v in args.other_workloads.split(",") if v)
            others, t_box = {}, time.perf_counter()
            A.close()
            for name in names:
                if time.perf_counter() - t_box > 150.0:
                    others[name] = {"skipped": "time box of 150 s spent on the shapes before it"}
                    continue
                try:
                    others[name] = measure_other_workload(name, dev)
                except Exception as e:
                    others[name] = {"error": repr(e)}
            out["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline and not device_built:
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

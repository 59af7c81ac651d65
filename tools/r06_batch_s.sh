#!/bin/bash
# round 6, batch s: the driver's bench command, the tests touched by the advisor items
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_driver_style.json 2> $OUT/r06_bench_driver_style.err; echo "bench rc $? ${SECONDS}s"
python3 - <<PY
import json
d = json.loads(open("$OUT/r06_bench_driver_style.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline value", round(d["value"], 1), "ms_per_step", d["ms_per_step"], "kernel_us", r["kernel_us"], "ratio", d["ms_per_step"] * 1e3 / r["kernel_us"], "frac", r["frac"], "dict_off", r.get("frac_value_dict_off"), "box", r.get("box_mode"))
for k, v in d.get("other_workloads", {}).items():
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("kernel_us", "frac", "wrong_rows", "frac_value_dict_off", "skipped", "error")})
c = d["cpu_baseline"]
print("cpu", {k: c.get(k) for k in ("value", "cores", "kind", "baseline_threads", "quota_limited", "spread")})
PY
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "phase_clocks or graph or panels or power" > $OUT/r06_pytest_subset.log 2>&1; echo "pytest rc $? ${SECONDS}s"; tail -3 $OUT/r06_pytest_subset.log

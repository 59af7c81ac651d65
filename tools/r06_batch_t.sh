#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2 3; do
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --other-workloads none > $OUT/r06_bench_t$i.json 2> $OUT/r06_bench_t$i.err
python3 - <<PY
import json
d = json.loads(open("$OUT/r06_bench_t$i.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step", round(d["ms_per_step"]*1e3,2), "kernel_us", round(r["kernel_us"],2), "ratio", round(d["ms_per_step"] * 1e3 / r["kernel_us"],3), "event_us/step", round(d["event_ms_per_step_rank0"]*1e3,2), d["timed_region_host_clock_us"])
PY
done

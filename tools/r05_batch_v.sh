#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 1200 python3 bench.py > $OUT/r05_bench.json 2> $OUT/r05_bench.err; echo "bench: ${SECONDS}s"
python3 - <<PY
import json
d = json.loads(open("$OUT/r05_bench.json").read().strip().splitlines()[-1])
c = d["cpu_baseline"]; print(d["value"], d["roofline"]["frac"], d["roofline"]["kernel_us"])
print({k: c[k] for k in ("value", "cores", "spread", "ms_per_step")}); print(c["per_thread_count"]); print(c["host_state"]); print(c.get("zeroing_inside_timer"))
PY

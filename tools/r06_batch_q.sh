#!/bin/bash
# round 6, batch q: gang kernel with 4 / 6 / 8 groups of gathers in flight per wavefront
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
run() { # workload, tag, CVR_DEBUG, extra args
  CVR_DEBUG="$3" timeout 600 python3 bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none $4 > $OUT/r06_q_$1_$2.json 2> $OUT/r06_q_$1_$2.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/r06_q_$1_$2.json").read().strip().splitlines()[-1])
    print("$1 %-28s kernel_us %7.1f frac %.4f wrong %d S %d chunks %d wpb %d" % ("$2", d["roofline"]["kernel_us"], d["roofline"]["frac"], d["verdict_wrong_rows"], d["config"]["steps_per_chunk"], d["config"]["chunks_rank0"], d["config"]["waves_per_workgroup"]), flush=True)
except Exception as e:
    print("$1 $2 no result:", e); print(open("$OUT/r06_q_$1_$2.err").read()[-800:])
PY
}
for w in livejournal orkut wikitalk; do
  run $w depth4 "" ""
  run $w depth6 "gang_depth=6" ""
  run $w depth8 "gang_depth=8" ""
  run $w depth4b "" ""
done

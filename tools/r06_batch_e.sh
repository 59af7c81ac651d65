#!/bin/bash
# round 6, batch e: gang chunks, launch parameters (helper wavefronts, pacing, sweep direction, non-temporal stream) and chunk lengths
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
run() { # workload, tag, CVR_DEBUG, extra args
  CVR_DEBUG="$3" timeout 600 python3 bench.py --workload $1 --steps 60 --warmup 10 --no-cpu-baseline --other-workloads none $4 > $OUT/r06_e_$1_$2.json 2> $OUT/r06_e_$1_$2.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/r06_e_$1_$2.json").read().strip().splitlines()[-1])
    print("$1 %-28s kernel_us %7.1f frac %.4f wrong %d S %d chunks %d" % ("$2", d["roofline"]["kernel_us"], d["roofline"]["frac"], d["verdict_wrong_rows"], d["config"]["steps_per_chunk"], d["config"]["chunks_rank0"]), flush=True)
except Exception as e:
    print("$1 $2 no result:", e); print(open("$OUT/r06_e_$1_$2.err").read()[-800:])
PY
}
for w in livejournal orkut; do
  run $w default "" ""
  run $w helpers0 "ilv_helpers=0" ""
  run $w helpers1 "ilv_helpers=1" ""
  run $w ahead12 "ilv_ahead=12" ""
  run $w ahead48 "ilv_ahead=48" ""
  run $w flip0 "ilv_flip=0" ""
  run $w nt0 "ilv_stream_nt=0" ""
  run $w perline1 "ilv_per_line=1" ""
done
run livejournal S576 "" "--steps-per-chunk 576"
run livejournal S508 "" "--steps-per-chunk 508"
run livejournal S400 "" "--steps-per-chunk 400"
run livejournal S320 "" "--steps-per-chunk 320"
run orkut S576 "" "--steps-per-chunk 576"
run orkut S444 "" "--steps-per-chunk 444"
echo "all ${SECONDS}s"

#!/bin/bash
# round 6, batch r: com-Orkut shape, combine pass batches and one helper wavefront per chunk, three runs each
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
run() { # workload, tag, CVR_DEBUG, extra args
  CVR_DEBUG="$3" timeout 600 python3 bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none $4 > $OUT/r06_r_$1_$2.json 2> $OUT/r06_r_$1_$2.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/r06_r_$1_$2.json").read().strip().splitlines()[-1])
    print("$1 %-28s kernel_us %7.1f frac %.4f wrong %d copy %d" % ("$2", d["roofline"]["kernel_us"], d["roofline"]["frac"], d["verdict_wrong_rows"], d["roofline"]["copy_kernel_gbs"]), flush=True)
except Exception as e:
    print("$1 $2 no result:", e); print(open("$OUT/r06_r_$1_$2.err").read()[-800:])
PY
}
for i in 1 2 3; do
  run orkut default$i "" ""
  run orkut batch8_$i "combine_batch=8" ""
  run orkut helpers1_$i "ilv_helpers=1" ""
  run orkut batch8h1_$i "combine_batch=8,ilv_helpers=1" ""
done
for i in 1 2; do
  run livejournal default$i "" ""
  run livejournal batch8_$i "combine_batch=8" ""
  run livejournal batch16_$i "combine_batch=16" ""
done

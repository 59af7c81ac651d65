#!/bin/bash
# round 6, batch h: the combine pass inside the panel kernel -- parity first, then the shapes with and without it
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 900 python3 tools/r06_gang_check.py > $OUT/r06_gang_check_fused.log 2>&1; echo "check rc $? ${SECONDS}s"; grep -v " same$" $OUT/r06_gang_check_fused.log | tail -12
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gang or panels or interleaved or image_cache or full_size" > $OUT/r06_pytest_fused.log 2>&1; echo "pytest rc $? ${SECONDS}s"; tail -6 $OUT/r06_pytest_fused.log
run() { # workload, tag, CVR_DEBUG, extra args
  CVR_DEBUG="$3" timeout 600 python3 bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none $4 > $OUT/r06_h_$1_$2.json 2> $OUT/r06_h_$1_$2.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/r06_h_$1_$2.json").read().strip().splitlines()[-1])
    print("$1 %-28s kernel_us %7.1f frac %.4f wrong %d S %d chunks %d panels %d cut %d" % ("$2", d["roofline"]["kernel_us"], d["roofline"]["frac"], d["verdict_wrong_rows"], d["config"]["steps_per_chunk"], d["config"]["chunks_rank0"], d["config"]["col_panels"], d["config"]["rows_cut_rank0"]), flush=True)
except Exception as e:
    print("$1 $2 no result:", e); print(open("$OUT/r06_h_$1_$2.err").read()[-800:])
PY
}
for w in livejournal orkut; do
  run $w fused "" ""
  run $w nofuse "no_fuse" ""
  run $w fused2 "" ""
done
echo "all ${SECONDS}s"

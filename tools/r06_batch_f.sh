#!/bin/bash
# round 6, batch f: the GPU suite's gang tests + the parity check script, panel counts with gang chunks
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 900 python3 tools/r06_gang_check.py > $OUT/r06_gang_check.log 2>&1; echo "check rc $? ${SECONDS}s"; grep -v " same$" $OUT/r06_gang_check.log | tail -12
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gang or interleaved or image_cache" > $OUT/r06_pytest_gang.log 2>&1; echo "pytest rc $? ${SECONDS}s"; tail -15 $OUT/r06_pytest_gang.log
timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "interleaved" > $OUT/r06_pytest_fuzz.log 2>&1; echo "fuzz rc $? ${SECONDS}s"; tail -8 $OUT/r06_pytest_fuzz.log
run() { # workload, tag, CVR_DEBUG, extra args
  CVR_DEBUG="$3" timeout 600 python3 bench.py --workload $1 --steps 60 --warmup 10 --no-cpu-baseline --other-workloads none $4 > $OUT/r06_f_$1_$2.json 2> $OUT/r06_f_$1_$2.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/r06_f_$1_$2.json").read().strip().splitlines()[-1])
    print("$1 %-28s kernel_us %7.1f frac %.4f wrong %d S %d chunks %d panels %d t_pre %.2f ms" % ("$2", d["roofline"]["kernel_us"], d["roofline"]["frac"], d["verdict_wrong_rows"], d["config"]["steps_per_chunk"], d["config"]["chunks_rank0"], d["config"]["col_panels"], d["preprocess"]["warm"]["t_pre_s"] * 1e3), flush=True)
except Exception as e:
    print("$1 $2 no result:", e); print(open("$OUT/r06_f_$1_$2.err").read()[-800:])
PY
}
run livejournal p8 "ilv_helpers=0" "--col-panels 8 --interleave 1"
run livejournal p12 "ilv_helpers=0" "--col-panels 12 --interleave 1"
run livejournal p16 "ilv_helpers=0" "--col-panels 16 --interleave 1"
run livejournal p24 "ilv_helpers=0" "--col-panels 24 --interleave 1"
run livejournal p32 "ilv_helpers=0" "--col-panels 32 --interleave 1"
run orkut p8 "ilv_helpers=0" "--col-panels 8 --interleave 1"
run orkut p12 "ilv_helpers=0" "--col-panels 12 --interleave 1"
run orkut p16 "ilv_helpers=0" "--col-panels 16 --interleave 1"
run wikitalk default "" ""
run wikitalk gang4 "" "--col-panels 8 --interleave 1"
echo "all ${SECONDS}s"

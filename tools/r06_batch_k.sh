#!/bin/bash
# round 6, batch k: wiki-Talk chunk lengths; the hold-out shapes with gang chunks in the rules; preprocessing phases of the soc-LiveJournal1 shape
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
run() { # workload, tag, CVR_DEBUG, extra args
  CVR_DEBUG="$3" timeout 600 python3 bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none $4 > $OUT/r06_k_$1_$2.json 2> $OUT/r06_k_$1_$2.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/r06_k_$1_$2.json").read().strip().splitlines()[-1])
    print("$1 %-28s kernel_us %7.1f frac %.4f wrong %d S %d chunks %d panels %d wpb %d" % ("$2", d["roofline"]["kernel_us"], d["roofline"]["frac"], d["verdict_wrong_rows"], d["config"]["steps_per_chunk"], d["config"]["chunks_rank0"], d["config"]["col_panels"], d["config"]["waves_per_workgroup"]), flush=True)
except Exception as e:
    print("$1 $2 no result:", e); print(open("$OUT/r06_k_$1_$2.err").read()[-800:])
PY
}
run wikitalk default "" ""
for S in 24 32 44 64 88 128; do run wikitalk S$S "" "--steps-per-chunk $S"; done
for S in 32 64; do run wikitalk S${S}_p16 "" "--steps-per-chunk $S --col-panels 16 --interleave 1"; done
echo "wikitalk ${SECONDS}s"
( timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r06_cvr_vs_csr_livejournal.log 2>&1; grep -E "T_pre|total|spmv_us|plan|convert|I_pre" $OUT/r06_cvr_vs_csr_livejournal.log | head -12
echo "compare ${SECONDS}s"
timeout 2400 python3 tools/holdout.py > $OUT/r06_holdout.log 2>&1; grep -E "^# |->" $OUT/r06_holdout.log | cut -c1-330
echo "all ${SECONDS}s"

#!/bin/bash
# round 6, batch d: gang chunks in the product -- parity on small cases, then the full-size shapes through bench.py
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 900 python3 tools/r06_gang_check.py > $OUT/r06_gang_check.log 2>&1; echo "check rc $? ${SECONDS}s"; grep -v " same$" $OUT/r06_gang_check.log | tail -30
for w in livejournal orkut wikitalk; do
  timeout 600 python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none > $OUT/r06_gang_$w.json 2> $OUT/r06_gang_$w.err; echo "$w rc $? ${SECONDS}s"
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/r06_gang_$w.json").read().strip().splitlines()[-1])
    print("$w", "kernel_us", round(d["roofline"]["kernel_us"], 1), "frac", round(d["roofline"]["frac"], 4), "wrong", d["verdict_wrong_rows"], "ms/step", round(d["ms_per_step"], 4), d["config"].get("layout"), "t_pre", d["preprocess"].get("t_pre_s"))
except Exception as e:
    print("$w", "no result:", e); print(open("$OUT/r06_gang_$w.err").read()[-1500:])
PY
done
CVR_DEBUG=no_gang timeout 600 python3 bench.py --workload livejournal --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none > $OUT/r06_nogang_livejournal.json 2> $OUT/r06_nogang_livejournal.err; python3 -c "
import json; d=json.loads(open('$OUT/r06_nogang_livejournal.json').read().strip().splitlines()[-1]); print('lj no gang', round(d['roofline']['kernel_us'],1), d['roofline']['frac'])"
echo "all ${SECONDS}s"

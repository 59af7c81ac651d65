#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel or image_cache or held_out or eight_ranks" > $OUT/r05_tests_w.txt 2>&1; grep -E "^E  |passed|failed|FAILED" $OUT/r05_tests_w.txt | head -10 | cut -c1-300
for w in livejournal orkut wikitalk; do python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', round(d['roofline']['kernel_us'], 2), round(d['roofline']['frac'], 4), d['verdict_wrong_rows'])
"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lj_trace_w3 -- python3 $R/bench.py --workload livejournal --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none > $OUT/lj_trace_w3.json 2>/dev/null
grep -E "spmv_ilv|combine" $OUT/lj_trace_w3/*/*kernel_stats.csv | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-160

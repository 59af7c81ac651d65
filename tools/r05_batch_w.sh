#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_y.txt 2>&1; grep -E "passed|failed" $OUT/r05_gpu_suite_y.txt | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for w in wikitalk livejournal; do python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', round(d['roofline']['kernel_us'], 2), round(d['roofline']['frac'], 4), d['verdict_wrong_rows'])
"; done

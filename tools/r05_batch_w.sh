#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for w in livejournal orkut; do for k in x combine_batch=13 combine_batch=14 x combine_batch=13; do CVR_DEBUG=$k python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', '$k', round(d['roofline']['kernel_us'], 2), round(d['roofline']['frac'], 4), d['verdict_wrong_rows'])
"; done; done | tee $OUT/r05_dense_combine_threads.log

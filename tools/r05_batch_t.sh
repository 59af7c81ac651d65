#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
HOLDOUT_LOG=$OUT/r05_holdout_sparse.log timeout 1500 python3 tools/holdout.py forum_sparse bipartite_sparse wikitalk_x2 > /dev/null 2>&1; grep -E "^#|automatic|wavefronts|->" $OUT/r05_holdout_sparse.log | cut -c1-230
for w in wikitalk livejournal; do python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', round(d['roofline']['kernel_us'], 2), round(d['roofline']['frac'], 4), d['verdict_wrong_rows'], d['config'].get('steps_per_chunk'), d['config'].get('waves_per_workgroup'), 't_pre_ms', d.get('t_pre_ms'))
"; done
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interleav or ilv or panel" > $OUT/r05_tests_t.txt 2>&1; tail -2 $OUT/r05_tests_t.txt

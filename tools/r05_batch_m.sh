#!/bin/bash
# tools/r05_batch_m.sh -- round 5: packed interleaved stream (4 bytes per slot): parity, then the shapes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or interleaved" > $OUT/r05_packed_tests.txt 2>&1; tail -15 $OUT/r05_packed_tests.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_m.txt 2>&1; tail -3 $OUT/r05_gpu_suite_m.txt
bash tools/final_numbers.sh r05m "livejournal orkut wikitalk"
for w in livejournal orkut; do CVR_DEBUG=no_pack32 python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w unpacked:', round(d['roofline']['kernel_us'], 2), round(d['roofline']['frac'], 4), d['verdict_wrong_rows'])
"; done

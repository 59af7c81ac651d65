#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
cp cvr_amd/libcvr_amd.so /tmp/plain.so; cp cvr_amd/libcvr_amd_A.so /tmp/A.so
run() { python3 bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', '$2', round(d['roofline']['kernel_us'], 3), round(d['roofline']['frac'], 4), d['verdict_wrong_rows'])
"; }
for rep in 1 2; do for w in rmat22 banded3.5e6 rmat20; do for v in plain A; do cp /tmp/$v.so cvr_amd/libcvr_amd.so; run $w $v; done; done; done | tee $OUT/r05_hub_nt_store.log
cp /tmp/plain.so cvr_amd/libcvr_amd.so
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_y.txt 2>&1; grep -E "passed|failed" $OUT/r05_gpu_suite_y.txt | tail -2

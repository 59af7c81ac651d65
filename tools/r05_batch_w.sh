#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
cp cvr_amd/libcvr_amd.so /tmp/plain.so; cp cvr_amd/libcvr_amd_A.so /tmp/A.so; cp cvr_amd/libcvr_amd_B.so /tmp/B.so
run() { python3 bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', '$2', round(d['roofline']['kernel_us'], 2), round(d['roofline']['frac'], 4), d['verdict_wrong_rows'])
"; }
for rep in 1 2; do for v in plain A; do cp /tmp/$v.so cvr_amd/libcvr_amd.so; run webgoogle $v; done; for v in plain B; do cp /tmp/$v.so cvr_amd/libcvr_amd.so; run rmat22 $v; done; done | tee $OUT/r05_policy_probe.log
cp /tmp/plain.so cvr_amd/libcvr_amd.so

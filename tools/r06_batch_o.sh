#!/bin/bash
# round 6, batch o: road hold-out under the short-row rule, the GPU suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 1200 python3 tools/holdout.py road citation > $OUT/r06_holdout_road.log 2>&1; grep -E "^# [a-z_0-9]+  |max regret|automatic  |plain  |plain S=16" $OUT/r06_holdout_road.log | cut -c1-170
echo "holdout ${SECONDS}s"
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/r06_pytest_gpu.log 2>&1; echo "pytest rc $? ${SECONDS}s"; tail -12 $OUT/r06_pytest_gpu.log

#!/bin/bash
# round 6, batch n: the hold-out sweep under the final rules, the GPU suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
HOLDOUT_LOG=$OUT/r06_holdout_lines.log timeout 2400 python3 tools/holdout.py > $OUT/r06_holdout.log 2>&1; grep -E "^# [a-z_0-9]+  |max regret" $OUT/r06_holdout.log
echo "holdout ${SECONDS}s"
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/r06_pytest_gpu.log 2>&1; echo "pytest rc $? ${SECONDS}s"; tail -12 $OUT/r06_pytest_gpu.log

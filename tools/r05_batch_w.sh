#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
rm -f $OUT/r05_holdout_end2.log
HOLDOUT_LOG=$OUT/r05_holdout_end2.log timeout 2400 python3 tools/holdout.py > /dev/null 2>&1; grep -E "^# " $OUT/r05_holdout_end2.log | tail -15 | cut -c1-200
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_y.txt 2>&1; grep -E "passed|failed" $OUT/r05_gpu_suite_y.txt | tail -2

#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1200 python3 tools/holdout.py citation road rmat21b webgoogle_real > $OUT/r06_holdout_single.log 2>&1; grep -E "^# [a-z_0-9]+:|1 image|automatic  |plain  |max regret" $OUT/r06_holdout_single.log | cut -c1-170

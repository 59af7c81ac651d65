#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 900 python3 tools/r05_eight_ranks_debug.py ) > $OUT/r05_eight_ranks_debug.log 2>&1; cat $OUT/r05_eight_ranks_debug.log | grep -v Gloo

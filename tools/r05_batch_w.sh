#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python3 tools/r05_convert_probe2.py 2>&1 | grep -E "^---|ilv_clocks|convert" | tee $OUT/r05_convert_probe2.log

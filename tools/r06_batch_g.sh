#!/bin/bash
# round 6, batch g: the whole GPU suite with gang chunks in the tree
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/r06_pytest_gpu.log 2>&1; echo "pytest rc $? ${SECONDS}s"; tail -25 $OUT/r06_pytest_gpu.log

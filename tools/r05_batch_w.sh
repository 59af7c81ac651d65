#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python3 tools/r05_one_generation.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_one_generation.log | cut -c1-900

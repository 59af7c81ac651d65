#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python3 tools/r05_sparse_combine_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_sparse_combine_threads2.log | cut -c1-300

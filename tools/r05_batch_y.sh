#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( CVR_DEBUG=create_timing=1 timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r05_create_timing_livejournal_y.log 2>&1; grep -E "\"plan\"|hub_selection|\"total\"|convert_device|preprocess_wall" $OUT/r05_create_timing_livejournal_y.log
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_y.txt 2>&1; grep -E "passed|failed" $OUT/r05_gpu_suite_y.txt | tail -2

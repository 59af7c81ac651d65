#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel or held_out or interleav or image_cache or eight_ranks" > $OUT/r05_tests_w.txt 2>&1; grep -E "^E  |passed|failed|FAILED" $OUT/r05_tests_w.txt | head -20 | cut -c1-300
for i in 1 2 3; do ( CVR_DEBUG=create_timing=1 timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r05_create_timing_livejournal_w.log 2>&1; grep "cvr_create\]" $OUT/r05_create_timing_livejournal_w.log | tail -14 | sed -n 5,9p; grep -E "\"plan\"|hub_selection|\"total\"|convert_device" $OUT/r05_create_timing_livejournal_w.log | head -4; done

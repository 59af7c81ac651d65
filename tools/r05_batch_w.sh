#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interleav or ilv" > $OUT/r05_tests_w.txt 2>&1; grep -E "^E  |passed|failed|FAILED" $OUT/r05_tests_w.txt | head -10 | cut -c1-300
( CVR_DEBUG=ilv_clocks timeout 900 python3 tools/compare_csr.py livejournal ) 2>&1 | grep -E "ilv_clocks|convert_device|\"total\"|result_ok" | head -5

#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2 3 4; do timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_ranks or two_ranks" > $OUT/r05_eight_ranks_test_q$i.txt 2>&1; tail -1 $OUT/r05_eight_ranks_test_q$i.txt; grep -E "^E  " $OUT/r05_eight_ranks_test_q$i.txt | head -3; done
( CVR_DEBUG=create_timing=1 timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r05_create_timing_livejournal.log 2>&1; grep "cvr_create\]" $OUT/r05_create_timing_livejournal.log | tail -40

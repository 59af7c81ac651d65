#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( CVR_DEBUG=create_timing=1 timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r05_create_timing_livejournal_u.log 2>&1; grep "cvr_create\]" $OUT/r05_create_timing_livejournal_u.log | tail -14 | head -9; grep -E "\"plan\"|hub_selection|\"total\"|convert_device" $OUT/r05_create_timing_livejournal_u.log
rm -f $OUT/r05_holdout_end.log
HOLDOUT_LOG=$OUT/r05_holdout_end.log timeout 2400 python3 tools/holdout.py > /dev/null 2>&1; grep -E "^# " $OUT/r05_holdout_end.log | tail -16 | cut -c1-200
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interleav or ilv or panel or held_out" > $OUT/r05_tests_u.txt 2>&1; tail -2 $OUT/r05_tests_u.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pre_trace_lj_u -- python3 $R/tools/compare_csr.py livejournal > $OUT/pre_trace_lj_u.log 2>&1
grep -E "est_count|hub_count|hub_share|ilv_chunk" $OUT/pre_trace_lj_u/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200

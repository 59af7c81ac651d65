#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_ranks or two_ranks" > $OUT/r05_eight_ranks_test_o$i.txt 2>&1; tail -1 $OUT/r05_eight_ranks_test_o$i.txt; grep -E "^E  " $OUT/r05_eight_ranks_test_o$i.txt | head -3; done
( timeout 900 python3 tools/r05_eight_ranks_debug.py ) 2>&1 | grep -v Gloo | grep -E "slice|one handle|verdict|rank |x as" > $OUT/r05_eight_ranks_debug_o.log; cat $OUT/r05_eight_ranks_debug_o.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_o.txt 2>&1; tail -3 $OUT/r05_gpu_suite_o.txt
( timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r05_cvr_vs_csr_livejournal.log 2>&1; tail -5 $OUT/r05_cvr_vs_csr_livejournal.log | cut -c1-400

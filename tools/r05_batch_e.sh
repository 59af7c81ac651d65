#!/bin/bash
# tools/r05_batch_e.sh -- round 5: helper wavefronts with both halves of a line, sweep direction alternating between SpMVs, chunks of up to 576 steps (com-Orkut shape),
# the panel rule's pairs criterion on the citation-like hold-out shape
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interleaved" > $OUT/r05_gpu_ilv_tests_e.txt 2>&1; tail -3 $OUT/r05_gpu_ilv_tests_e.txt
( timeout 1200 python3 tools/helper_probe.py livejournal "0,16,1" "0,16,1,dbg_ilv_flip=1" "3,24,2" "3,24,2,dbg_ilv_flip=1" "3,12,2" "3,16,2" "3,32,2" "3,48,2" "1,24,2" "2,24,2" "2,16,2" "3,24,2,steps_per_chunk=444,col_panels=16,interleave=1,waves_per_block=3" ) > $OUT/r05_helper_probe2_lj.log 2>&1; cat $OUT/r05_helper_probe2_lj.log
( timeout 1200 python3 tools/helper_probe.py orkut "0,16,1" "3,24,2" "3,24,2,dbg_ilv_flip=1" "0,16,1,dbg_ilv_flip=1" "3,24,2,steps_per_chunk=508,col_panels=8,interleave=1" "0,16,1,steps_per_chunk=508,col_panels=8,interleave=1") > $OUT/r05_helper_probe2_orkut.log 2>&1; cat $OUT/r05_helper_probe2_orkut.log
( timeout 600 python3 tools/helper_probe.py wikitalk "0,16,1" "3,24,2" "3,8,2" "0,16,1,dbg_ilv_flip=1" ) > $OUT/r05_helper_probe2_wikitalk.log 2>&1; cat $OUT/r05_helper_probe2_wikitalk.log
HOLDOUT_LOG=$OUT/r05_holdout_citation.log timeout 600 python3 tools/holdout.py citation > /dev/null 2>&1; tail -4 $OUT/r05_holdout_citation.log

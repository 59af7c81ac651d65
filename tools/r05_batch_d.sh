#!/bin/bash
# tools/r05_batch_d.sh -- round 5: GPU suite on the kernels with the ring from v40, helper wavefronts (scalar prefetch of the stream) on the three
# interleaved shapes, phase clocks of the headline kernel, hold-out shapes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_d.txt 2>&1; tail -5 $OUT/r05_gpu_suite_d.txt
( timeout 900 python3 tools/helper_probe.py livejournal "0,16,1" "1,16,1" "2,16,1" "3,16,1" "3,12,1" "3,20,1" "3,24,1" "3,32,1" "3,16,2" "2,24,2" "3,24,2" "3,16,1,col_panels=32,interleave=1" "3,24,1,col_panels=32,interleave=1" ) > $OUT/r05_helper_probe_lj.log 2>&1; cat $OUT/r05_helper_probe_lj.log
( timeout 900 python3 tools/helper_probe.py orkut "0,16,1" "3,16,1" "3,24,1" "3,24,2" ) > $OUT/r05_helper_probe_orkut.log 2>&1; cat $OUT/r05_helper_probe_orkut.log
( timeout 600 python3 tools/helper_probe.py wikitalk "0,16,1" "3,16,1" "3,24,1" ) > $OUT/r05_helper_probe_wikitalk.log 2>&1; cat $OUT/r05_helper_probe_wikitalk.log
( timeout 300 python3 tools/phase_clocks.py webgoogle ) > $OUT/r05_phase_clocks_webgoogle.txt 2>&1; cat $OUT/r05_phase_clocks_webgoogle.txt
HOLDOUT_LOG=$OUT/r05_holdout.log timeout 1500 python3 tools/holdout.py > $OUT/r05_holdout_stdout.log 2>&1; tail -15 $OUT/r05_holdout.log

#!/bin/bash
# tools/r05_batch_f.sh -- round 5: the scalar path's rate, the headline kernel's phase clocks after the prologue change, every shape through bench.py, hold-out shapes
# with the panel rule's pairs criterion, GPU suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 tools/ubench/smem_rate ) > $OUT/r05_smem_rate_ubench.log 2>&1; cat $OUT/r05_smem_rate_ubench.log
( timeout 300 python3 tools/phase_clocks.py webgoogle ) > $OUT/r05_phase_clocks_webgoogle_b.txt 2>&1; cat $OUT/r05_phase_clocks_webgoogle_b.txt
bash tools/final_numbers.sh r05f "webgoogle livejournal orkut wikitalk rmat22"
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_f.txt 2>&1; tail -3 $OUT/r05_gpu_suite_f.txt
HOLDOUT_LOG=$OUT/r05_holdout_b.log timeout 1500 python3 tools/holdout.py > /dev/null 2>&1; tail -13 $OUT/r05_holdout_b.log

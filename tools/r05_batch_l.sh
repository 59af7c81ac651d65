#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_ranks" > $OUT/r05_eight_ranks_test_l$i.txt 2>&1; tail -1 $OUT/r05_eight_ranks_test_l$i.txt; grep -E "^E  " $OUT/r05_eight_ranks_test_l$i.txt | head -3; done

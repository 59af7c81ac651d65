#!/bin/bash
# tools/r05_batch_c.sh -- round 5: (a) is the interleaved kernel held back by the matrix stream's HBM misses in the L1's miss queue?  The prototype with every
# chunk streaming one of 8 chunk images of its panel (L2-resident stream, timing only); (b) the GPU suite on the rebuilt library
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( for m in 0 8; do echo "### SAME_STREAM=$m"; SAME_STREAM=$m timeout 600 python3 tools/sorted_probe.py lj "4096 4 400 16 0 9 1" "5000 4 444 16 0 9 1" "5000 4 444 16 0 9 1 0 1"; done ) > $OUT/r05_same_stream_probe.log 2>&1
grep "RESULT\|###\|^## " $OUT/r05_same_stream_probe.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_c.txt 2>&1; tail -5 $OUT/r05_gpu_suite_c.txt

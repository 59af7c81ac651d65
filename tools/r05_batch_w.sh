#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
: > $OUT/r05_wpb_helpers2.log
timeout 900 python3 tools/helper_probe.py livejournal 2,24,2 3,24,2,waves_per_block=3,col_panels=16,interleave=1 4,24,2,waves_per_block=3,col_panels=16,interleave=1 4,32,2,waves_per_block=3,col_panels=16,interleave=1 6,24,2,waves_per_block=2,col_panels=16,interleave=1 2,24,2 3,24,2,waves_per_block=3,col_panels=16,interleave=1 2>&1 | grep -v amdgpu.ids >> $OUT/r05_wpb_helpers2.log
timeout 900 python3 tools/helper_probe.py orkut 2,24,2 3,24,2,waves_per_block=3,col_panels=8,interleave=1 4,24,2,waves_per_block=3,col_panels=8,interleave=1 2,24,2 2>&1 | grep -v amdgpu.ids >> $OUT/r05_wpb_helpers2.log
cut -c1-220 $OUT/r05_wpb_helpers2.log

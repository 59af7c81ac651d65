#!/bin/bash
# round 6, batch l: wiki-Talk shape, workgroup shapes with gang chunks; the sparse hold-out shapes under the new rule
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 900 python3 tools/layout_probe.py wikitalk "col_panels=8,interleave=1,waves_per_block=4,gang=1" "col_panels=8,interleave=1,waves_per_block=8,gang=1" "col_panels=8,interleave=1,waves_per_block=8,gang=1,steps_per_chunk=44" "col_panels=8,interleave=1,waves_per_block=8,gang=1,steps_per_chunk=64" "col_panels=8,interleave=1,waves_per_block=4,gang=1,steps_per_chunk=64" "col_panels=8,interleave=1,waves_per_block=4,gang=1,steps_per_chunk=128" "col_panels=8,interleave=1,waves_per_block=2,gang=1" "col_panels=4,interleave=1,waves_per_block=4,gang=1" "col_panels=8,interleave=1,waves_per_block=2,gang=0" > $OUT/r06_wikitalk_layouts.log 2>&1; cat $OUT/r06_wikitalk_layouts.log | cut -c1-220
echo "probe ${SECONDS}s"
timeout 1200 python3 tools/holdout.py forum_sparse bipartite_sparse wikitalk_x2 citation > $OUT/r06_holdout_sparse.log 2>&1; grep -E "^# [a-z_0-9]+  |max regret" $OUT/r06_holdout_sparse.log
echo "all ${SECONDS}s"

#!/bin/bash
# round 6, batch p: wiki-Talk shape in the prototype: what the LDS additions cost (hot rows: lanes of one instruction adding to one accumulator)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 900 python3 tools/sorted_probe.py wikitalk \
  "5000 2 176 8 0 9 1" \
  "5000 2 176 8 0 9 1 1" \
  "EXE=sorted_spmv_s0 TOK_U=2 1100 4 92 8 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=2 1100 4 92 8 5 9 1 1" \
  "EXE=sorted_spmv_s0 TOK_U=2 2200 2 176 8 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=2 2200 2 176 8 5 9 1 1" \
  ) > $OUT/r06_wikitalk_noadd_probe.log 2>&1
grep -E "^##|RESULT|^# f64" $OUT/r06_wikitalk_noadd_probe.log | cut -c1-220

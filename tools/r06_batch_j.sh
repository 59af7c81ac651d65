#!/bin/bash
# round 6, batch j: prototype, 4-byte packed slots (offset 13 | tag 15 | code 4) against 5-byte slots, token units of 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 1500 python3 tools/sorted_probe.py lj \
  "EXE=sorted_spmv_nt TOK_U=2 4800 4 2100 16 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4800 4 2100 16 6 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4800 4 2100 16 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4800 4 2100 16 6 9 1" \
  ) > $OUT/r06_pack4_probe_lj.log 2>&1
grep -E "^##|RESULT|rerun|mode" $OUT/r06_pack4_probe_lj.log
( timeout 1500 python3 tools/sorted_probe.py orkut \
  "EXE=sorted_spmv_nt TOK_U=2 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4096 4 2100 8 6 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4096 4 2100 8 6 9 1" \
  ) > $OUT/r06_pack4_probe_orkut.log 2>&1
grep -E "^##|RESULT|rerun|mode" $OUT/r06_pack4_probe_orkut.log

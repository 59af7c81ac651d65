#!/bin/bash
# round 6, batch a: the prototype's token-ordered shared chunk (modes 4 / 5) against private chunks, soc-LiveJournal1 and com-Orkut shapes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
( timeout 1500 python3 tools/sorted_probe.py lj \
  "5000 4 444 16 0 9 1" \
  "4096 4 1800 16 1 9 1" \
  "TOK_U=1 4096 4 1800 16 4 9 1" \
  "TOK_U=2 4096 4 1800 16 4 9 1" \
  "TOK_U=4 4096 4 1800 16 4 9 1" \
  "TOK_U=1 4096 4 1800 16 5 9 1" \
  "TOK_U=2 4096 4 1800 16 5 9 1" \
  "TOK_U=4 4096 4 1800 16 5 9 1" \
  "TOK_U=2 4800 4 2100 16 5 9 1" \
  "EXE=sorted_spmv_nt 5000 4 444 16 0 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4096 4 1800 16 5 9 1" \
  "TOK_U=2 2048 8 1800 16 5 9 1" \
  "SAME_STREAM=8 TOK_U=2 4096 4 1800 16 5 9 1" \
  ) > $OUT/r06_token_probe_lj.log 2>&1
echo "lj: ${SECONDS}s"; grep -E "^##|RESULT|rerun|mode" $OUT/r06_token_probe_lj.log
( timeout 1500 python3 tools/sorted_probe.py orkut \
  "5000 4 516 8 0 9 1" \
  "TOK_U=2 4096 4 2100 8 5 9 1" \
  "TOK_U=1 4096 4 2100 8 5 9 1" \
  "TOK_U=4 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=2 4096 4 2100 8 5 9 1" \
  "TOK_U=2 4096 4 2100 16 5 9 1" \
  ) > $OUT/r06_token_probe_orkut.log 2>&1
echo "orkut: ${SECONDS}s"; grep -E "^##|RESULT|rerun|mode" $OUT/r06_token_probe_orkut.log

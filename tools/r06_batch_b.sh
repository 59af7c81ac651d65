#!/bin/bash
# round 6, batch b: token-ordered shared chunk with nothing but the additions inside the token's hold
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
( timeout 1500 python3 tools/sorted_probe.py lj \
  "TOK_U=1 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=1 4096 4 1800 16 5 9 1" \
  "TOK_U=2 4096 4 1800 16 5 9 1" \
  "TOK_U=1 4096 4 1800 16 4 9 1" \
  "TOK_U=1 4800 4 2100 16 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=1 4096 4 1800 16 5 9 1" \
  "TOK_U=1 2048 8 1800 16 5 9 1" \
  "TOK_U=1 4096 4 1800 8 5 9 1" \
  "SAME_STREAM=8 TOK_U=1 4096 4 1800 16 5 9 1" \
  ) > $OUT/r06_token_probe_lj_b.log 2>&1
echo "lj: ${SECONDS}s"; grep -E "^##|RESULT|rerun" $OUT/r06_token_probe_lj_b.log
( timeout 1500 python3 tools/sorted_probe.py orkut \
  "TOK_U=1 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=1 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=1 4096 4 2100 8 5 9 1" \
  "TOK_U=1 4800 4 2400 8 5 9 1" \
  "TOK_U=1 4096 4 2100 8 4 9 1" \
  ) > $OUT/r06_token_probe_orkut_b.log 2>&1
echo "orkut: ${SECONDS}s"; grep -E "^##|RESULT|rerun" $OUT/r06_token_probe_orkut_b.log
( timeout 900 python3 tools/sorted_probe.py wikitalk \
  "5000 4 444 8 0 9 1" \
  "5000 2 176 8 0 9 1" \
  "TOK_U=1 4096 4 1800 8 5 9 1" \
  "TOK_U=1 4800 4 1800 8 5 9 1" \
  "TOK_U=1 4800 4 1800 4 5 9 1" \
  ) > $OUT/r06_token_probe_wikitalk_b.log 2>&1
echo "wikitalk: ${SECONDS}s"; grep -E "^##|RESULT|rerun" $OUT/r06_token_probe_wikitalk_b.log

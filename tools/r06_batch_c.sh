#!/bin/bash
# round 6, batch c: token units of 1-8 groups whose products wait in registers (only the additions inside the token's hold)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
( timeout 1500 python3 tools/sorted_probe.py lj \
  "EXE=sorted_spmv_s0 TOK_U=1 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=2 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=4 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=8 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv TOK_U=4 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=4 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=4 4800 4 2100 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=4 2400 8 2100 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=2 2400 8 2100 16 5 9 1" \
  "EXE=sorted_spmv_s0 SAME_STREAM=8 TOK_U=4 4096 4 1800 16 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=4 4096 4 1800 24 5 9 1" \
  ) > $OUT/r06_token_probe_lj_c.log 2>&1
echo "lj: ${SECONDS}s"; grep -E "^##|RESULT|rerun" $OUT/r06_token_probe_lj_c.log
( timeout 1500 python3 tools/sorted_probe.py orkut \
  "EXE=sorted_spmv_s0 TOK_U=2 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=4 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=8 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_nt TOK_U=4 4096 4 2100 8 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=4 2400 8 2100 8 5 9 1" \
  "EXE=sorted_spmv_s0 TOK_U=4 4096 4 1600 8 5 9 1" \
  ) > $OUT/r06_token_probe_orkut_c.log 2>&1
echo "orkut: ${SECONDS}s"; grep -E "^##|RESULT|rerun" $OUT/r06_token_probe_orkut_c.log

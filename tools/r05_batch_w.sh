#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
bash tools/rank_emulation.sh rmat26 8 > /dev/null 2>&1; python3 -c "
import json; d=json.load(open('gpurun_out/rank_emulation_rmat26.json')); print('rmat26', [round(x['kernel_us']) for x in d['per_rank']], d['all_rows_checked_wrong'], [x['col_panels'] for x in d['per_rank']])"
bash tools/rank_emulation.sh banded28e6 8 > /dev/null 2>&1; python3 -c "
import json; d=json.load(open('gpurun_out/rank_emulation_banded28e6.json')); print('banded28e6', [round(x['kernel_us']) for x in d['per_rank']], d['all_rows_checked_wrong'])"
echo "${SECONDS}s"

#!/bin/bash
# tools/r03_lj_sweep2.sh -- soc-LiveJournal1 shape: panel counts and chunk lengths around the automatic choice (16 panels, S = 32)
R=$GRAFT_REPO_ROOT; cd $R
for cfg in "-1 0" "16 24" "16 28" "16 36" "16 40" "8 32" "24 32" "16 32"; do
  set -- $cfg
  timeout 200 python3 bench.py --workload livejournal --col-panels $1 --steps-per-chunk $2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('panels asked $1 S asked $2: panels %s S %s  us/step %.1f kernel %.1f wrong %s' % (d['config']['col_panels'], d['config']['steps_per_chunk'], d['ms_per_step']*1e3, d['roofline']['kernel_us'], d['verdict_wrong_rows']))"
done

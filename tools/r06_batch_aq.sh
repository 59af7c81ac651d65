#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for i in 1 2 3; do
SECONDS=0; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --other-workloads none > $OUT/r06_x.json 2>/dev/null
python3 - <<PY
import json
d = json.loads(open("$OUT/r06_x.json").read().strip().splitlines()[-1])
print('K=20 headline kernel_us', round(d['roofline']['kernel_us'],2), round(d['roofline']['frac'],4), 'ms_per_step', round(d['ms_per_step']*1e3,2), 'median single', round(d['roofline']['kernel_us_median_single_launches'],2))
PY
done
timeout 900 python3 bench.py --gpus 1 --no-cpu-baseline --other-workloads none > $OUT/r06_x.json 2>/dev/null
python3 - <<PY
import json
d = json.loads(open("$OUT/r06_x.json").read().strip().splitlines()[-1])
print('K=1000 headline kernel_us', round(d['roofline']['kernel_us'],2), round(d['roofline']['frac'],4), 'ms_per_step', round(d['ms_per_step']*1e3,2))
PY

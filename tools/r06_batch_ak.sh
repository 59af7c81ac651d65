#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
SECONDS=0
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_driver_style.json 2> $OUT/r06_bench_driver_style.err; echo "bench rc $? ${SECONDS}s"
python3 - <<PY
import json
d = json.loads(open("$OUT/r06_bench_driver_style.json").read().strip().splitlines()[-1])
print('headline', round(d['roofline']['kernel_us'],2), round(d['roofline']['frac'],4), d['ms_per_step'], round(d['value'],1), d['roofline'].get('box_mode'))
for k,w in d['other_workloads'].items():
    print(k, round(w['kernel_us'],2), round(w['frac'],4), w.get('frac_value_dict_off'), w['wrong_rows'])
PY

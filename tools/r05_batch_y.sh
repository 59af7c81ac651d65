#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/r05_gpu_suite_y.txt 2>&1; grep -E "passed|failed" $OUT/r05_gpu_suite_y.txt | tail -2
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SECONDS=0; python3 bench.py > $OUT/r05_bench_last.json 2>/dev/null; echo "bench ${SECONDS}s"; python3 -c "
import json; d=json.loads(open('$OUT/r05_bench_last.json').read().strip().splitlines()[-1]); print(d['metric'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['spread'], {k: round(v['frac'],3) for k,v in d['other_workloads'].items()})"

#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lds_or_with_atomics" > $OUT/r05_tests_w.txt 2>&1; grep -E "^E  |passed|failed|FAILED" $OUT/r05_tests_w.txt | head -20 | cut -c1-300
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pre_trace_lj_w -- python3 $R/tools/compare_csr.py livejournal > $OUT/pre_trace_lj_w.log 2>&1
python3 -c "
import csv,glob
f=glob.glob('$OUT/pre_trace_lj_w/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('est_','hub_')): print(round(float(r['AverageNs'])/1e3,1), r['Calls'], r['Name'][:80])
"
grep -E "\"plan\"|hub_selection|\"total\"" $OUT/pre_trace_lj_w.log | head -4

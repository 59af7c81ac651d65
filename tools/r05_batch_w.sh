#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2 3; do ( CVR_DEBUG=create_timing=1 timeout 900 python3 tools/compare_csr.py livejournal ) 2>&1 | grep -E "panel split|\"total\"|result_ok" | tail -3 | head -2; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pre_trace_lj_w4 -- python3 $R/tools/compare_csr.py livejournal > $OUT/pre_trace_lj_w4.log 2>&1
python3 -c "
import csv,glob
f=glob.glob('$OUT/pre_trace_lj_w4/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('part_','split_')): print(round(float(r['AverageNs'])/1e3,1), r['Calls'], r['Name'][:80])
"

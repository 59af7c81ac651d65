#!/bin/bash
# preprocessing of the soc-LiveJournal1 shape: phases of cvr_create on the host clock, kernel durations under rocprofv3, the amortisation report
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( CVR_DEBUG=create_timing=1 timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r05_create_timing_livejournal_r.log 2>&1; grep "cvr_create\]" $OUT/r05_create_timing_livejournal_r.log | tail -14; grep -E "plan:|hub_selection|dict_scan|convert_device|total:" $OUT/r05_create_timing_livejournal_r.log
( CVR_DEBUG=ilv_clocks timeout 900 python3 tools/compare_csr.py livejournal ) 2>&1 | grep ilv_clocks | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pre_trace_lj -- python3 $R/tools/compare_csr.py livejournal > $OUT/pre_trace_lj.log 2>&1
cp $OUT/pre_trace_lj/*/*kernel_stats.csv $OUT/r05_pre_kernel_stats_livejournal.csv 2>/dev/null
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/r05_pre_kernel_stats_livejournal.csv")))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("rocsparse", "csr_vector", "spmv_ilv", "combine_kernel")): continue
    print(f'{float(r["AverageNs"])/1e3:10.1f} us x {r["Calls"]:>5}  {n[:110]}')
PY
cd $R
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel or dict or split or holdout or held_out" > $OUT/r05_tests_r.txt 2>&1; tail -2 $OUT/r05_tests_r.txt

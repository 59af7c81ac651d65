#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
: > $OUT/r05_combine_tables2.log
timeout 900 python3 tools/helper_probe.py livejournal 2,24,2,dbg_combine_batch=4 2,24,2,dbg_combine_batch=4,dbg_combine_mul=8 2,24,2,dbg_combine_batch=8,dbg_combine_mul=8 2>&1 | grep -v amdgpu.ids >> $OUT/r05_combine_tables2.log
timeout 900 python3 tools/helper_probe.py orkut 2,24,2,dbg_combine_batch=4 2,24,2,dbg_combine_batch=4,dbg_combine_mul=8 2,24,2,dbg_combine_batch=8,dbg_combine_mul=8 2>&1 | grep -v amdgpu.ids >> $OUT/r05_combine_tables2.log
cut -c1-200 $OUT/r05_combine_tables2.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lj_trace_w -- python3 $R/bench.py --workload livejournal --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none > $OUT/lj_trace_w.json 2>/dev/null
grep -E "spmv_ilv|combine" $OUT/lj_trace_w/*/*kernel_stats.csv | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-160

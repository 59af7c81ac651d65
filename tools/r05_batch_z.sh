#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp
export TMPDIR=/tmp
for w in wikitalk; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $R/bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none > $OUT/trace_$w.json 2>/dev/null
grep -E "spmv_ilv|combine|fixup" $OUT/trace_$w/*/*kernel_stats.csv | awk -F'",' '{print substr($1,1,90), $2}' | cut -c1-200
done

#!/bin/bash
# tools/r05_batch_h.sh -- round 5: GPU suite after the revert, the 8-rank test three times, the combine pass with eight row blocks per workgroup (wiki-Talk shape)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_bench_eight_ranks_on_one_device > $OUT/r05_gpu_suite_h.txt 2>&1; tail -3 $OUT/r05_gpu_suite_h.txt
for i in 1 2 3; do timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_ranks" > $OUT/r05_eight_ranks_test_$i.txt 2>&1; tail -2 $OUT/r05_eight_ranks_test_$i.txt; grep -E "AssertionError|assert " $OUT/r05_eight_ranks_test_$i.txt | head -5; done
( timeout 600 python3 tools/helper_probe.py wikitalk "0,16,1,dbg_combine_mul=1,dbg_combine_batch=4" "0,16,1,dbg_combine_mul=8,dbg_combine_batch=4" "0,16,1,dbg_combine_mul=8,dbg_combine_batch=8" "0,16,1" "0,16,1,waves_per_block=2,col_panels=8,interleave=1" "0,16,1,waves_per_block=2,col_panels=8,interleave=1,dbg_combine_mul=1,dbg_combine_batch=4" ) > $OUT/r05_wikitalk_combine.log 2>&1; cat $OUT/r05_wikitalk_combine.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_trace_wt -- python3 $R/bench.py --workload wikitalk --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none > $OUT/r05_trace_wt.json 2> $OUT/r05_trace_wt.err
cp $OUT/r05_trace_wt/*/*kernel_stats.csv $OUT/r05_wikitalk_kernel_stats_h.csv 2>/dev/null; rm -rf $OUT/r05_trace_wt; head -4 $OUT/r05_wikitalk_kernel_stats_h.csv | cut -c1-150

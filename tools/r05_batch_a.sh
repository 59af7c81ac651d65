#!/bin/bash
# tools/r05_batch_a.sh -- round 5, first look: (a) the interleaved prototype with FEWER, LONGER private chunks per workgroup (2 x 8-10 k rows,
# 1 x 16-20 k rows instead of 4 x 4 k), (b) the product with waves_per_block = 2 / 1, (c) per-kernel times of the other shapes' SpMV steps
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 900 python3 tools/sorted_probe.py lj "4096 4 400 16 0 9 1" "8192 2 800 16 0 9 1" "10000 2 1000 16 0 9 1" "16384 1 1600 16 0 9 1" "19900 1 2000 16 0 9 1" "8192 2 800 32 0 9 1" "5000 4 500 32 0 9 1" "16384 1 1600 32 0 9 1" ) > $OUT/r05_big_chunk_prototype.log 2>&1
( timeout 900 python3 tools/layout_probe.py livejournal "waves_per_block=2" "waves_per_block=2,steps_per_chunk=508" "waves_per_block=1,steps_per_chunk=508" "waves_per_block=3" "waves_per_block=2,col_panels=32" ) > $OUT/r05_wpb_product_lj.log 2>&1
cd /tmp
for w in wikitalk orkut rmat22; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_trace_$w -- python3 $R/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --other-workloads none > $OUT/r05_trace_$w.json 2> $OUT/r05_trace_$w.err
  cp $OUT/r05_trace_$w/*/*kernel_stats.csv $OUT/r05_${w}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/r05_trace_$w
done
head -5 $OUT/r05_*_kernel_stats.csv
tail -30 $OUT/r05_big_chunk_prototype.log
cat $OUT/r05_wpb_product_lj.log

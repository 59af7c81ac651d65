#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2 3 4; do
  CVR_BENCH_DEBUG_SKIP="uploaded,buffers, stream,warmup,kernel_time" CVR_BENCH_ONE_DEVICE=1 CVR_BENCH_NO_TUNE=1 CVR_BENCH_DEBUG_DUMP=1 timeout 600 python3 bench.py --gpus 8 --steps 10 --warmup 2 --workload rmat20 --no-cpu-baseline --dump-y /tmp/y8.npy 2>&1 | grep -E "x differs|x ok at|verdict_wrong" | cut -c1-600 > $OUT/r05_eight_ranks_debug_q$i.log; grep -c "x ok" $OUT/r05_eight_ranks_debug_q$i.log; grep "x differs" $OUT/r05_eight_ranks_debug_q$i.log
done

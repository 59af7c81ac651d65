#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pair_list or panel or image_cache or power" > $OUT/r05_tests_w.txt 2>&1; tail -5 $OUT/r05_tests_w.txt
: > $OUT/r05_sparse_combine.log
timeout 900 python3 tools/helper_probe.py wikitalk 0,16,1 0,16,1,dbg_no_sparse_combine=1 2>&1 | grep -v amdgpu.ids >> $OUT/r05_sparse_combine.log
HOLDOUT_LOG=$OUT/r05_sparse_combine.log timeout 900 python3 tools/holdout.py forum_sparse wikitalk_x2 > /dev/null 2>&1
grep -E "^#|helpers|automatic|2 wavefronts" $OUT/r05_sparse_combine.log | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_wikitalk2 -- python3 $R/bench.py --workload wikitalk --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none > $OUT/trace_wikitalk2.json 2>/dev/null
grep -E "spmv_ilv|combine|fixup" $OUT/trace_wikitalk2/*/*kernel_stats.csv | awk -F'",' '{print substr($1,1,90), $2}' | cut -c1-200

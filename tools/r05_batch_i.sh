#!/bin/bash
# tools/r05_batch_i.sh -- round 5: the 8-rank path with the whole gathered vector verified in the run; helper wavefronts of the hub-table kernel (R-MAT);
# the interleaved kernel's prologue; GPU suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2; do ( CVR_BENCH_ONE_DEVICE=1 CVR_BENCH_NO_TUNE=1 timeout 900 python3 bench.py --gpus 8 --steps 10 --warmup 2 --workload rmat20 --no-cpu-baseline --dump-y /tmp/y8.npy ) > $OUT/r05_eight_ranks_rmat20_$i.json 2> $OUT/r05_eight_ranks_rmat20_$i.err; python3 -c "
import json
d = json.loads([l for l in open('$OUT/r05_eight_ranks_rmat20_$i.json') if l.startswith('{')][-1])
print({k: d.get(k) for k in ('n_gpus', 'verdict_wrong_rows', 'gathered_slices_differing_between_ranks', 'gather_impl')})
"; done
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_ranks" > $OUT/r05_eight_ranks_test_i.txt 2>&1; tail -2 $OUT/r05_eight_ranks_test_i.txt; grep -E "^E " $OUT/r05_eight_ranks_test_i.txt | head -5
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_bench_eight_ranks_on_one_device > $OUT/r05_gpu_suite_i.txt 2>&1; tail -3 $OUT/r05_gpu_suite_i.txt
( timeout 900 python3 tools/helper_probe.py rmat22 "0,16,1" "0,16,1,dbg_hub_helpers=1,dbg_hub_ahead=3" "0,16,1,dbg_hub_helpers=1,dbg_hub_ahead=4" "0,16,1,dbg_hub_helpers=1,dbg_hub_ahead=6" "0,16,1,dbg_hub_helpers=1,dbg_hub_ahead=8" "0,16,1,dbg_hub_helpers=1,dbg_hub_ahead=12" ) > $OUT/r05_rmat_helpers.log 2>&1; cat $OUT/r05_rmat_helpers.log
bash tools/final_numbers.sh r05i "webgoogle livejournal orkut wikitalk rmat22"
( timeout 600 python3 tools/helper_probe.py wikitalk "0,16,1" "0,16,1,waves_per_block=2,col_panels=8,interleave=1" "0,16,1,waves_per_block=1,col_panels=8,interleave=1" ) > $OUT/r05_wikitalk_wpb.log 2>&1; cat $OUT/r05_wikitalk_wpb.log

#!/bin/bash
# tools/r05_batch_g.sh -- round 5: phase rotation + window barrier where the walk reaches the window (headline), scalar path rate, 8 ranks on one device,
# combine pass variants on the wiki-Talk shape, R-MAT with an L2-resident stream (ceiling of a stream prefetch)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_bench_eight_ranks_on_one_device > $OUT/r05_gpu_suite_g.txt 2>&1; tail -3 $OUT/r05_gpu_suite_g.txt
( timeout 300 python3 tools/phase_clocks.py webgoogle ) > $OUT/r05_phase_clocks_webgoogle_c.txt 2>&1; cat $OUT/r05_phase_clocks_webgoogle_c.txt
( CVR_DEBUG=no_phase_rot timeout 300 python3 tools/phase_clocks.py webgoogle ) > $OUT/r05_phase_clocks_webgoogle_c_norot.txt 2>&1; head -12 $OUT/r05_phase_clocks_webgoogle_c_norot.txt
for i in 1 2; do python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('webgoogle bench:', d['roofline']['kernel_us'], d['roofline']['frac'], d['ms_per_step'], d['roofline'].get('kernel_us_value_dict_off'))
"; done
CVR_DEBUG=no_phase_rot python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('webgoogle bench no_phase_rot:', d['roofline']['kernel_us'], d['roofline']['frac'], d['ms_per_step'])
"
( timeout 300 tools/ubench/smem_rate ) > $OUT/r05_smem_rate_ubench.log 2>&1; cat $OUT/r05_smem_rate_ubench.log
( CVR_BENCH_ONE_DEVICE=1 CVR_BENCH_NO_TUNE=1 timeout 900 python3 bench.py --gpus 8 --steps 10 --warmup 2 --workload rmat20 --no-cpu-baseline --dump-y /tmp/y8.npy ) > $OUT/r05_eight_ranks_rmat20.json 2> $OUT/r05_eight_ranks_rmat20.err; tail -5 $OUT/r05_eight_ranks_rmat20.err; python3 -c "
import json
d = json.loads([l for l in open('$OUT/r05_eight_ranks_rmat20.json') if l.startswith('{')][-1])
print({k: d.get(k) for k in ('n_gpus', 'verdict_wrong_rows', 'gathered_slices_differing_between_ranks', 'gather_impl')}, d['config'].get('rows_per_gpu'))
"
( timeout 600 python3 tools/helper_probe.py wikitalk "0,16,1" "0,16,1,dbg_combine_batch=8" "0,16,1,waves_per_block=1,col_panels=8,interleave=1" "0,16,1,waves_per_block=2,col_panels=8,interleave=1" "0,16,1,waves_per_block=1,col_panels=8,interleave=1,dbg_combine_batch=8" ) > $OUT/r05_wikitalk_variants.log 2>&1; cat $OUT/r05_wikitalk_variants.log
( timeout 600 python3 tools/helper_probe.py rmat22 "0,16,1" "0,16,1,dbg_stream_mod=64" "0,16,1,dbg_stream_mod=16" ) > $OUT/r05_rmat_stream_mod.log 2>&1; cat $OUT/r05_rmat_stream_mod.log

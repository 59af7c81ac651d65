#!/bin/bash
# tools/final_numbers.sh TAG ['workload ...'] -- every shape of DESIGN.md section 5.1 through bench.py on one GPU (kernel time over 200 launches, wrong rows), one JSON line each
# (copy_kernel_gbs: the 1-GiB copy kernel on the same box in the same process -- boxes of the pool differ by 15 % in it, and the stream-bound shapes with them)
# -> gpurun_out/TAG_final_numbers.log
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TAG}_final_numbers.log; [ -n "$2" ] || : > $OUT
for w in ${2:-webgoogle livejournal orkut wikitalk rmat22 rmat24 rmat26 banded3.5e6 banded28e6}; do
  extra=""; steps=200; case $w in banded28e6|rmat26) steps=50;; esac          # (the two 8-GPU configurations whole on one GPU)
  python3 $R/bench.py --workload $w --steps $steps --warmup 20 --no-cpu-baseline --other-workloads none $extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; c = d['config']
    print(json.dumps({'workload': '$w $extra'.strip(), 'kernel_us': round(r['kernel_us'], 2), 'frac': round(r['frac'], 4), 'ms_per_step': round(d['ms_per_step'], 5), 'dtype': d['dtype'], 'nnz': c['rank_nnz'], 'wrong_rows': d['verdict_wrong_rows'],
                      'copy_kernel_gbs': round(r.get('copy_kernel_gbs') or 0), 'layout': {k: c.get(k) for k in ('steps_per_chunk', 'waves_per_workgroup', 'col_panels', 'col_phases', 'x_window_values', 'value_dictionary_entries')}, 'kernel': r['kernel']}))
" >> $OUT
done
cat $OUT

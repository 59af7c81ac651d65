#!/bin/bash
# tools/pmc_convert.sh -- SQ counters of the preprocessing kernels of the web-Google shape (tools/wg_create_once.py): instruction
# mix and wait cycles of convert_kernel / seg_fill_kernel.  Two short --pmc passes, each time-boxed.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_convert; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/tools/wg_create_once.py > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
        if not any(k in n for k in ("convert_kernel", "seg_fill", "seg_count", "window_kernel")): continue
        a = agg[n][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for n, d in agg.items():
    print(n, {k: round(v[0] / v[1]) for k, v in sorted(d.items())})
PY
rm -rf $OUT

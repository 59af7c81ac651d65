#!/bin/bash
# tools/pmc_tlb.sh -- address-translation counters of the SpMV kernel on the web-Google shape (is the gather paying for TLB misses?)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_tlb; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum --output-format csv -d $OUT/p1 -- python3 $R/tools/sweep.py webgoogle --S 56 --swz 1 --nt 0 --iters 3 --warmup 0 > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmv_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({k: round(v[0] / v[1]) for k, v in sorted(agg.items())})
PY
tail -3 $OUT/err.txt

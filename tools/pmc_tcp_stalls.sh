#!/bin/bash
# tools/pmc_tcp_stalls.sh -- what the CU's vector L1 (TCP) waits on during the SpMV kernel (web-Google shape): stall cycle
# counters of the tag lookup, the miss queue, the return path and the address / data paths to the texture addresser
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_tcp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/tools/sweep.py webgoogle --S 56 --swz 1 --nt 0 --iters 3 --warmup 0 > /dev/null 2> $OUT/err$i.txt
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
dur = []
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmv_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in sorted(agg.items()):
    print("%-44s %14.0f per launch  = %8.0f per CU" % (k, v[0] / v[1], v[0] / v[1] / 256))
PY

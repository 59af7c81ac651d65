#!/bin/bash
# tools/pmc_groups.sh TAG "sweep.py args" -- memory-pipeline counters of the SpMV kernel, one rocprofv3 run per group
TAG=$1; shift
ARGS=${@:-webgoogle --S 32 --swz 1 --nt 0 --win 0 --iters 3 --warmup 0}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_LATENCY_FIFO_FULL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/tools/sweep.py $ARGS > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmv_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(agg):
        line = f"{k:42s} {agg[k][0] / agg[k][1]:16.1f}"
        print(line); o.write(line + "\n")
PY

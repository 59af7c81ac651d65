#!/bin/bash
# tools/pmc.sh OUTDIR -- PMC passes for the SpMV kernel (each counter group in its own rocprofv3 run, no tracing
# domains besides the kernel trace; MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run on the GPU box.  Every pass is time-boxed:
# the TA_* group hung a pass for 40 minutes on a large matrix in round 2 and is no longer collected.
OUT=${1:-gpurun_out/pmc}; shift
ARGS=${@:-webgoogle --S 32 --swz 1 --nt 0 --iters 20}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/p$i -- python3 $R/tools/sweep.py $ARGS > $R/$OUT.p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$R/$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmv_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("$R/$OUT/summary.txt", "w") as o:
    for k in sorted(agg):
        line = f"{k:45s} mean/launch {agg[k][0] / agg[k][1]:16.1f}   launches {agg[k][1]}"
        print(line); o.write(line + "\n")
PY

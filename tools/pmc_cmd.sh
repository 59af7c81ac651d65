#!/bin/bash
# tools/pmc_cmd.sh TAG KERNEL_SUBSTRING python3 script.py args...  -- per-launch means of the gather counters (L1->L2 requests and
# their latency, L2 hits / misses, fabric reads, HBM-side bytes) for the kernels whose name contains KERNEL_SUBSTRING, each counter
# group in a pass of its own with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Summary: gpurun_out/TAG/pmc_summary.json
TAG=$1; KSUB=$2; shift; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1 )
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0.0, 0])
names = collections.Counter()
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$KSUB" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        names[r["Kernel_Name"].split("(")[0][-60:]] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "$KSUB" not in r["Kernel_Name"]: continue
        d = dur[r["Kernel_Name"].split("(")[0][-60:]]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
out = {k: v[0] / v[1] for k, v in sorted(agg.items())}
out["launches_per_counter"] = {k: v[1] for k, v in sorted(agg.items())}
out["kernel_us_under_the_profiler"] = {k: v[0] / v[1] for k, v in dur.items()}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:      # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies a 128-byte read at 64 (MI355X_MICROARCH.md, HBM)
    out["hbm_bytes_per_launch_corrected"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
if "TCC_HIT_sum" in out: out["l2_hit_rate"] = out["TCC_HIT_sum"] / max(out["TCC_HIT_sum"] + out["TCC_MISS_sum"], 1)
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print("$TAG", json.dumps(out, indent=1))
PY

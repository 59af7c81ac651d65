#!/bin/bash
# tools/profile_bench.sh TAG [bench.py arguments] -- on the GPU box: rocprofv3 kernel-trace stats of the bench command, then the
# HBM traffic counters in separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; gpurun
# refuses --pmc combined with tracing domains other than the kernel trace).  Summaries -> gpurun_out/TAG/; the PMC summary
# records the configuration it was taken on (kernel, S, chunk count, image bytes, workgroup layout), which bench.py compares
# with the configuration it times before it reports the figure as roofline.traffic.
TAG=${1:-r02}; shift
ARGS="$@"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CVR_BENCH_NO_DICT_OFF_RUN=1        # one SpMV configuration per profiled process
CMD="python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --other-workloads none $ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_traced.json 2> $OUT/trace.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  SECONDS=0; timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-workloads none $ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc pass $i ($grp): ${SECONDS}s"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmv_kernel" not in r["Kernel_Name"] and "spmv_seg_kernel" not in r["Kernel_Name"] and "spmv_ilv_kernel" not in r["Kernel_Name"] and "spmv_gang_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
s = {k: agg[k][0] / agg[k][1] for k in agg}
# FETCH_SIZE / WRITE_SIZE come in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B: double it
# (MI355X_MICROARCH.md, HBM section)
if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
    s["hbm_bytes_per_launch_corrected"] = (2.0 * s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024.0
    s["hbm_bytes_per_launch_raw"] = (s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024.0
if "TCC_HIT_sum" in s and "TCC_MISS_sum" in s: s["l2_hit_rate"] = s["TCC_HIT_sum"] / max(s["TCC_HIT_sum"] + s["TCC_MISS_sum"], 1.0)
s["launches_per_counter"] = {k: agg[k][1] for k in agg}
try:      # the configuration the counters belong to, from the bench line of a PMC pass
    d = json.loads([l for l in open("$OUT/pmc1.json") if l.startswith("{")][-1])
    c = d["config"]
    s["config"] = {"kernel": d["roofline"]["kernel"], "steps_per_chunk": c["steps_per_chunk"], "nchunks": c["chunks_rank0"],
                   "image_bytes": d["roofline"]["streamed_bytes_per_launch"] - d["roofline"]["algorithmic_bytes_per_launch"] * 0,
                   "waves_per_block": c["waves_per_workgroup"], "col_phases": c["col_phases"], "x_window": c["x_window_values"],
                   "workload": c["workload"], "col_panels": c["col_panels"], "bench_args": "$ARGS"}
    s["config"]["image_bytes"] = d["image_bytes"]
except Exception as e:
    s["config_error"] = repr(e)
json.dump(s, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY

#!/bin/bash
# tools/trace_workload.sh WORKLOAD TAG -- rocprofv3 kernel-trace stats of bench.py on another workload (livejournal, banded<rows>):
# per-kernel durations of the multi-kernel (column panel) path.  Summary -> gpurun_out/TAG/kernel_stats.csv
W=${1:-livejournal}
TAG=${2:-trace_$W}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last full iteration: kernels between two consecutive combine kernels
    idx = [i for i, r in enumerate(rows) if "combine_kernel" in r["Kernel_Name"]]
    if len(idx) >= 2:
        a, b = idx[-2] + 1, idx[-1] + 1
        t0 = int(rows[a]["Start_Timestamp"])
        with open("$OUT/last_iteration.txt", "w") as o:
            for r in rows[a:b]:
                s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
                o.write("%9.1f us .. %9.1f us  (%7.1f us, gap before %5.1f)  %s  grid %s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, 0.0, r["Kernel_Name"][:60], r.get("Grid_Size", "")))
    break
PY
rm -rf $OUT/trace
head -12 $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/last_iteration.txt 2>/dev/null

#!/bin/bash
# tools/r03_pre_timeline.sh -- on the GPU box: HIP-API + kernel timeline of the last (warm) cvr_create + cvr_preprocess of the web-Google
# shape (tools/wg_create_once.py), merged and printed relative to the layout probe's start.  Output: gpurun_out/r03_pre_timeline.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pre_timeline; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT -- python3 $R/tools/wg_create_once.py > $OUT/run.log 2> $OUT/run.err
python3 - <<PY > $R/gpurun_out/r03_pre_timeline.txt
import csv, glob
ev = []
for f in glob.glob("$OUT/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:90] + "  grid %s wg %s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))))
for f in glob.glob("$OUT/*/*hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A", r["Function"]))
ev.sort()
probes = [e for e in ev if e[2] == "K" and "probe" in e[3]]
t0 = probes[-1][0]
print(open("$OUT/run.log").read())
print("# last create: times in us relative to the start of the layout probe kernel; K = kernel, A = HIP API call")
for s, e, k, n in ev:
    if s < t0 - 400e3 or s > t0 + 900e3: continue
    if k == "A" and e - s < 1500 and not any(w in n for w in ("Launch", "Synchronize", "Malloc", "Free")): continue
    print("%9.1f %9.1f %7.1f  %s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, n))
PY
tail -n 120 $R/gpurun_out/r03_pre_timeline.txt

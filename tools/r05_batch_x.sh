#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp
export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --stats --output-format csv -d $OUT/pre_hip_lj -- python3 $R/tools/compare_csr.py livejournal > $OUT/pre_hip_lj.log 2>&1
ls $OUT/pre_hip_lj/*/ | head
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/pre_hip_lj/*/*hip_api_stats.csv") + glob.glob("$OUT/pre_hip_lj/*/*hip_stats.csv"):
    print(f)
    for r in list(csv.DictReader(open(f)))[:25]:
        print(f'{r["Name"][:40]:40s} calls {r["Calls"]:>7} total {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e3:9.1f} us')
PY

"""tools/trace_timeline.py DIR PATTERN [BEFORE AFTER] -- kernels of a rocprofv3 --kernel-trace run around the LAST kernel whose name
contains PATTERN: start (us after it) and duration."""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 12
after = int(sys.argv[4]) if len(sys.argv) > 4 else 40
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[max(0, idx - before): idx + after]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))

"""tools/trace_summary.py DIR [T_FROM_FRACTION] -- kernels and HIP API calls of a rocprofv3 --kernel-trace --hip-trace run, summed over
the last part of the run (default: the second half, i.e. the second of two identical creates)."""
import csv, glob, sys, collections
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
def load(pat):
    f = glob.glob(d + "/*/*" + pat)
    return list(csv.DictReader(open(f[0]))) if f else []
k = load("kernel_trace.csv"); a = load("hip_api_trace.csv")
ts = [int(r["Start_Timestamp"]) for r in k + a]
t0, t1 = min(ts), max(ts)
cut = t0 + (t1 - t0) * frac
for rows, key, title in ((k, "Kernel_Name", "kernels"), (a, "Function", "HIP API")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if int(r["Start_Timestamp"]) < cut: continue
        x = agg[r[key].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]]; x[0] += 1; x[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"## {title} after {frac:.2f} of the run: total {sum(v[1] for v in agg.values())/1e3:.2f} ms")
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"  {n:70s} {v[0]:6d} calls {v[1]/1e3:9.3f} ms")

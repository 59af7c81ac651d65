#!/bin/bash
# tools/locality_report.sh [matrix] -- the MI355X counterpart of the reference's run_locality.sh (VTune L2 hit/miss of
# spmv_compute_kernel, solutions_for_comparison/run_locality.sh:39-55; paper section 7.4, Fig. 7): L2 hit rate,
# L1->L2 requests and fabric read requests of the CVR64 kernel and of the GPU CSR comparators, per launch.
MAT=${1:-webgoogle}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/locality_$MAT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  g=$(echo $grp | cut -c1-7)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$g -- python3 $R/tests/compare_csr.py $MAT 3 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "cvr64 spmv_kernel" if "spmv_kernel" in k else "csr_vector (own)" if "csr_vector_kernel" in k else "rocsparse " + k.split("(")[0][-40:] if "rocsparse" in k else None
        if name is None: continue
        a = agg[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"# $MAT: per-launch means")
print(f"{'kernel':58s} {'L2 hit %':>9s} {'L1->L2 reads':>13s} {'avg lat (clk)':>13s} {'fabric reads (128 B)':>20s}")
for name, c in sorted(agg.items()):
    m = {k: v[0] / v[1] for k, v in c.items()}
    if m.get("TCC_REQ_sum", 0) < 1000: continue
    hit = 100.0 * m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
    lat = m.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(m.get("TCP_TCC_READ_REQ_sum", 1), 1)
    print(f"{name:58s} {hit:9.1f} {m.get('TCP_TCC_READ_REQ_sum', 0):13.0f} {lat:13.0f} {m.get('TCC_EA0_RDREQ_sum', 0):20.0f}")
PY

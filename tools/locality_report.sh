#!/bin/bash
# tools/locality_report.sh [matrix] [iters] -- the MI355X counterpart of the reference's run_locality.sh (VTune L2 hit / miss of
# spmv_compute_kernel, /root/reference/solutions_for_comparison/run_locality.sh:39-55; paper section 7.4, Fig. 7): per SpMV, for the
# CVR64 kernels (every launch an SpMV is made of: panel rounds, combine, fix-up, hub compaction) and for the GPU CSR comparators
# (own CSR-vector kernel, rocSPARSE adaptive / row-split): L1->L2 read requests and their mean latency, L2 hit rate, and what goes
# past the L2 -- fabric read requests, those destined for DRAM, and the mean latency of a fabric read, which tells the Infinity
# Cache (MALL) from HBM (MI355X_MICROARCH.md: ~545 clk for an Infinity-Cache hit, ~900 for an HBM miss, idle chip).
# Counter groups in separate passes with --kernel-trace only.  Output: gpurun_out/locality_MATRIX.txt
MAT=${1:-webgoogle}; IT=${2:-20}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/locality_$MAT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  ( cd $R && timeout 420 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/compare_csr.py $MAT $IT > $OUT/p$i.json 2> $OUT/p$i.err )
done
python3 - <<PY > $R/gpurun_out/locality_$MAT.txt
import csv, glob, collections, json
it = $IT
def group(k):
    if "rocsparse" in k:
        return "rocSPARSE " + ("adaptive" if "adaptive" in k or "csrmvn_adaptive" in k else "row-split / general" if "csrmvn" in k else "other")
    if "csr_vector_kernel" in k: return "CSR-vector (own comparator)"
    for s in ("spmv_ilv_kernel", "spmv_seg_kernel", "spmv_kernel", "combine_kernel", "fixup", "hub_gather"):
        if s in k: return "CVR64 (all kernels of an SpMV)"
    return None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        g = group(r["Kernel_Name"])
        if g is None: continue
        agg[g][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[g][r["Counter_Name"]] += 1
dur = collections.defaultdict(float)
for f in glob.glob("$OUT/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        g = group(r["Kernel_Name"])
        if g: dur[g] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
try:
    rep = json.loads(open("$OUT/p1.json").read())
    lay = rep["cvr"]["layout"]
    print(f"# $MAT: {rep['rows']} rows, {rep['nnz']} non-zeros, fp64; CVR64 layout {lay}")
    print(f"# event-timed in the profiled process (counter collection serialises the launches: not the kernels' own times): CVR64 {rep['cvr']['spmv_us']:.1f} us; " + "; ".join(f"{k} {v.get('spmv_us', float('nan')):.1f} us" for k, v in rep["baselines"].items()))
except Exception as e:
    print("# $MAT (report line unavailable: %r)" % (e,))
# SpMVs executed per group: compare_csr runs 1 warm-up + 1 timed (cvr_spmv) + 20 + iters (bench) for CVR64, 20 + iters per comparator kernel
nsp = {"CVR64 (all kernels of an SpMV)": 22 + it}
print("# per SpMV (sums over all launches / number of SpMVs)")
nnz = rep["nnz"] if "rep" in dir() else 0
print("# requests / nnz: L1->L2 read requests of all kernels of an SpMV (x gathers, matrix stream, tables) per non-zero; L1 lines: the vector L1s' cache-line accesses (tag look-ups)")
print(f"{'kernels':34s} {'us (profiled)':>13s} {'L1->L2 reads':>13s} {'req / nnz':>10s} {'L1 lines':>12s} {'L1 hit %':>9s} {'lat (clk)':>10s} {'L2 hit %':>9s} {'fabric reads':>13s} {'to DRAM':>12s} {'fabric lat (clk)':>16s}")
for g, c in sorted(agg.items()):
    n = nsp.get(g, 20 + it)
    hit = 100.0 * c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
    lat = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(c.get("TCP_TCC_READ_REQ_sum", 1), 1)
    flat = c.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / max(c.get("TCC_EA0_RDREQ_sum", 1), 1)
    l1 = c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / n
    l1hit = 100.0 * (1.0 - (c.get("TCP_TCC_READ_REQ_sum", 0) / n) / l1) if l1 > 0 else float("nan")
    print(f"{g:34s} {dur[g] / n:13.1f} {c.get('TCP_TCC_READ_REQ_sum', 0) / n:13.0f} {c.get('TCP_TCC_READ_REQ_sum', 0) / n / max(nnz, 1):10.3f} {l1:12.0f} {l1hit:9.1f} {lat:10.0f} {hit:9.1f} {c.get('TCC_EA0_RDREQ_sum', 0) / n:13.0f} {c.get('TCC_EA0_RDREQ_DRAM_sum', 0) / n:12.0f} {flat:16.0f}")
PY
cat $R/gpurun_out/locality_$MAT.txt

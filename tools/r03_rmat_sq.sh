#!/bin/bash
# tools/r03_rmat_sq.sh -- on the GPU box: instruction-issue counters of the SpMV kernel on R-MAT-22 fp32 (hub table): is the general kernel's ~70
# vector instructions per step what it waits for?  Output: gpurun_out/r03_rmat22_sq.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rmat_sq; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/tools/rmat_reorder_probe.py 22 f32 > $OUT/p$i.log 2> $OUT/p$i.err
done
python3 - <<PY > $R/gpurun_out/r03_rmat22_sq.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmv_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("# R-MAT-22 fp32, hub table, general spmv_kernel: per launch (counter collection serialises kernels; 67.1 M non-zeros = 1.08 M wavefront steps)")
for k, v in sorted(agg.items()):
    print("%-28s %16.0f per launch   (%d launches)" % (k, v[0] / v[1], v[1]))
PY
cat $R/gpurun_out/r03_rmat22_sq.txt

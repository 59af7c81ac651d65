#!/bin/bash
# tools/r03_convert_clocks.sh -- on the GPU box: stage clocks (CVR_CONVERT_CLOCKS, CVR_SEG_CLOCKS) of the converter and the segment-table kernel
# on the web-Google shape (staged path, so that nothing else runs beside them), for the default converter and the LDS-staged one, and the
# kernels' durations from a kernel trace.  Output: gpurun_out/r03_convert_clocks.log
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_convert_clocks.log; cd $R
export CVR_NO_FUSED=1
{
echo "# converter with per-lane global loads (CVR_NO_DICT_CODES=1: the round-2 kernel), every 256th chunk"
CVR_NO_DICT_CODES=1 CVR_CONVERT_CLOCKS=1 python3 tools/wg_create_once.py 2>&1 | grep "\[convert\]" | tail -7
echo "# LDS-staged converter fed with the values (CVR_NO_DICT_CODES=1 CVR_CONVERT_LDS=1)"
CVR_NO_DICT_CODES=1 CVR_CONVERT_LDS=1 CVR_CONVERT_CLOCKS=1 python3 tools/wg_create_once.py 2>&1 | grep "\[convert\]" | tail -7
echo "# LDS-staged converter fed with dictionary codes (the default when the matrix has a dictionary)"
CVR_CONVERT_CLOCKS=1 python3 tools/wg_create_once.py 2>&1 | grep "\[convert\]" | tail -7
echo "# segment table (seg_scan_kernel), every 256th chunk"
CVR_SEG_CLOCKS=1 python3 tools/wg_create_once.py 2>&1 | grep "\[seg_scan\]" | tail -8
} > $OUT
cd /tmp; export TMPDIR=/tmp
for v in codes values_lds values_global; do
  unset CVR_CONVERT_LDS CVR_NO_DICT_CODES
  if [ $v = values_lds ]; then export CVR_CONVERT_LDS=1 CVR_NO_DICT_CODES=1; fi
  if [ $v = values_global ]; then export CVR_NO_DICT_CODES=1; fi
  rm -rf /tmp/cc_trace; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cc_trace -- python3 $R/tools/wg_create_once.py > /dev/null 2>&1
  echo "# kernel trace, converter fed with $v: name, calls, average ns" >> $OUT
  python3 - <<PY >> $OUT
import csv, glob
for f in glob.glob("/tmp/cc_trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("convert", "seg_", "window_kernel", "dict_codes")): print("%-60s %5s %10.0f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
done
cat $OUT

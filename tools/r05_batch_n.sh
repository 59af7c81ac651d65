#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for i in 1 2; do ( timeout 900 python3 tools/r05_eight_ranks_debug.py ) 2>&1 | grep -v Gloo | grep -E "slice 7|slice 1:|one handle|verdict" ; done
echo "--- torch imported first"
for i in 1 2; do ( DEBUG_TORCH_FIRST=1 timeout 900 python3 tools/r05_eight_ranks_debug.py ) 2>&1 | grep -v Gloo | grep -E "slice 7|slice 1:|one handle|verdict" ; done
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_bench_eight_ranks_on_one_device > $OUT/r05_gpu_suite_n.txt 2>&1; tail -3 $OUT/r05_gpu_suite_n.txt

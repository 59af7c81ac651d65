#!/bin/bash
# tools/r05_batch_j.sh -- round 5: are the R-MAT shards slices of the whole matrix on the GPU?  two against four wavefronts per workgroup where chunks come out short; GPU suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 600 python3 tools/r05_rmat_shard_check.py ) > $OUT/r05_rmat_shard_check.log 2>&1; cat $OUT/r05_rmat_shard_check.log
( timeout 1500 python3 tools/wpb_probe.py ) > $OUT/r05_wpb_probe.log 2>&1; cat $OUT/r05_wpb_probe.log
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_bench_eight_ranks_on_one_device > $OUT/r05_gpu_suite_j.txt 2>&1; tail -3 $OUT/r05_gpu_suite_j.txt

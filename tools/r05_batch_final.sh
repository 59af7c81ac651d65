#!/bin/bash
# round 5, end-of-round records: default bench line, rocprofv3 kernel stats + PMC passes of the same command (headline and soc-LiveJournal1 shape), every shape, amortisation
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 900 python3 bench.py > $OUT/r05_bench.json 2> $OUT/r05_bench.err; echo "bench: ${SECONDS}s"; cut -c1-400 $OUT/r05_bench.json
bash tools/profile_bench.sh r05_bench > $OUT/r05_profile_bench.log 2>&1; tail -5 $OUT/r05_profile_bench.log
bash tools/profile_bench.sh r05_bench_livejournal --workload livejournal > $OUT/r05_profile_bench_lj.log 2>&1; tail -3 $OUT/r05_profile_bench_lj.log
cd $R
bash tools/final_numbers.sh r05 > /dev/null 2>&1; cat $OUT/r05_final_numbers.log | cut -c1-200
( timeout 600 python3 tools/compare_csr.py webgoogle ) > $OUT/r05_cvr_vs_csr_webgoogle.log 2>&1; grep -E "total|spmv_us" $OUT/r05_cvr_vs_csr_webgoogle.log | head -6
( timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/r05_cvr_vs_csr_livejournal.log 2>&1; grep -E "total|spmv_us|\"plan\"|plan:" $OUT/r05_cvr_vs_csr_livejournal.log | head -8
echo "all: ${SECONDS}s"

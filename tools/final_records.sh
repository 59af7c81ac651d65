#!/bin/bash
# tools/final_records.sh TAG -- end-of-round records on the GPU box: the GPU suite, the default bench line, rocprofv3 kernel stats + PMC passes of the same command (headline, soc-LiveJournal1 and
# com-Orkut shapes), every shape, the hold-out sweep, amortisation
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
SECONDS=0
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $? ${SECONDS}s"; tail -4 $OUT/${TAG}_pytest_gpu.log
timeout 900 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench: ${SECONDS}s"; cut -c1-300 $OUT/${TAG}_bench.json
bash tools/profile_bench.sh ${TAG}_bench > $OUT/${TAG}_profile_bench.log 2>&1; tail -3 $OUT/${TAG}_profile_bench.log
bash tools/profile_bench.sh ${TAG}_bench_livejournal --workload livejournal > $OUT/${TAG}_profile_bench_lj.log 2>&1; tail -3 $OUT/${TAG}_profile_bench_lj.log
bash tools/profile_bench.sh ${TAG}_bench_orkut --workload orkut > $OUT/${TAG}_profile_bench_orkut.log 2>&1; tail -3 $OUT/${TAG}_profile_bench_orkut.log
bash tools/profile_bench.sh ${TAG}_bench_wikitalk --workload wikitalk > $OUT/${TAG}_profile_bench_wikitalk.log 2>&1; tail -3 $OUT/${TAG}_profile_bench_wikitalk.log
cd $R
echo "profiles: ${SECONDS}s"
bash tools/final_numbers.sh r06 > /dev/null 2>&1; cat $OUT/${TAG}_final_numbers.log | cut -c1-200
echo "final numbers: ${SECONDS}s"
( timeout 600 python3 tools/compare_csr.py webgoogle ) > $OUT/${TAG}_cvr_vs_csr_webgoogle.log 2>&1; grep -E "\"total\"|spmv_us" $OUT/${TAG}_cvr_vs_csr_webgoogle.log | head -6
( timeout 900 python3 tools/compare_csr.py livejournal ) > $OUT/${TAG}_cvr_vs_csr_livejournal.log 2>&1; grep -E "\"total\"|spmv_us|\"plan\"" $OUT/${TAG}_cvr_vs_csr_livejournal.log | head -8
timeout 2400 python3 tools/holdout.py > $OUT/${TAG}_holdout.log 2>&1; grep -E "^# [a-z_0-9]+  |max regret" $OUT/${TAG}_holdout.log
echo "all: ${SECONDS}s"

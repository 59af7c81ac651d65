#!/bin/bash
# round 6, batch i: rocprofv3 stats + PMC passes of the soc-LiveJournal1 and com-Orkut shapes with gang chunks
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_bench.sh r06_lj_gang --workload livejournal > gpurun_out/r06_profile_lj_gang.log 2>&1; tail -40 gpurun_out/r06_profile_lj_gang.log | head -60
head -5 gpurun_out/r06_lj_gang/kernel_stats.csv | cut -c1-200
bash tools/profile_bench.sh r06_orkut_gang --workload orkut > gpurun_out/r06_profile_orkut_gang.log 2>&1; tail -30 gpurun_out/r06_profile_orkut_gang.log | head -40
head -5 gpurun_out/r06_orkut_gang/kernel_stats.csv | cut -c1-200

# pacing of long chunks through the column phases (spmv_seg_kernel, CVR_PACE_LAG: 0 = off): soc-LiveJournal1 shape as two row bands
mkdir -p gpurun_out/r3h
for LAG in 0 1 2 3; do
  echo "## CVR_PACE_LAG=$LAG"
  CVR_PACE_LAG=$LAG python tools/band_probe.py livejournal --bands 2 --wpb 8 --phases 16,24,32 --pmax 8 --check 2>&1 | grep "^# wpb\|wrong [1-9]"
done 2>&1 | tee gpurun_out/r3h/pace_sweep.log

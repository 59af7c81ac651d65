# R-MAT-22 fp32 as ONE resident launch (row sums in LDS, column phases over x, pieces of 8, pacing): tools/band_probe.py
mkdir -p gpurun_out/r3i
for LAG in 0 2 3; do
  echo "## CVR_PACE_LAG=$LAG"
  CVR_PACE_LAG=$LAG python tools/band_probe.py rmat22 --bands 1 --wpb 8,16 --phases 8,16,24 --pmax 8 --check 2>&1 | grep "^# wpb\|wrong [1-9]\|Error\|error"
done 2>&1 | tee gpurun_out/r3i/rmat22_band.log

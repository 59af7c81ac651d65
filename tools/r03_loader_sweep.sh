# window loaders (spmv_seg_kernel LOADER): wavefronts that bring the LDS window in while the others start; CVR_WIN_GROUP = group in front of which they meet
mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_parity.py -x -q -k "phases or timed or window" 2>&1 | tail -4
for L in 0 1 2 4; do for GR in 0 1 2 3; do
  if [ $L = 0 ] && [ $GR != 0 ]; then continue; fi
  echo "loaders $L group $GR"
  CVR_WIN_LOADERS=$L CVR_WIN_GROUP=$GR python tools/sweep.py webgoogle --S 0 --swz 1 --win -1 --phases -1 --iters 1000 --check 2>&1 | grep -v "^#"
done; done 2>&1 | tee gpurun_out/r3d/loader_sweep.log | cut -c1-120

mkdir -p gpurun_out/r3n
for L in 0 2 4 6; do for GR in -1 0; do
  if [ $L = 0 ] && [ $GR != 0 ]; then continue; fi
  echo "loaders $L group $GR"
  CVR_WIN_LOADERS=$L CVR_WIN_GROUP=$GR python tools/sweep.py webgoogle --S 0 --swz 1 --win -1 --phases -1 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-100
  CVR_WIN_LOADERS=$L CVR_WIN_GROUP=$GR python tools/sweep.py webgoogle --S 0 --swz 1 --win -1 --phases -1 --iters 2000 --dict 0 2>&1 | grep -v "^#" | cut -c1-100
done; done 2>&1 | tee gpurun_out/r3n/loader_sweep2.log

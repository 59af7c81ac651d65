# headline shape, column-phase kernel with loaders: window size against the rows an accumulator stage can hold (LDS is shared)
mkdir -p gpurun_out/r3q
for W in 6144 8192 10240 12288 14336; do
  python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win $W --phases 12,16 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-200
done 2>&1 | tee gpurun_out/r3q/window_sizes.log
python tools/sweep.py webgoogle --S 44 --swz 1 --wpb 8 --win 8192,10240 --phases 12 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-200 | tee -a gpurun_out/r3q/window_sizes.log
python tools/sweep.py webgoogle --S 56 --swz 1 --wpb 6 --win 8192,12288 --phases 12 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-200 | tee -a gpurun_out/r3q/window_sizes.log

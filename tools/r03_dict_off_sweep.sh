# headline shape with full fp64 values in the stream (value_dict = 0): chunk length / workgroup shape / window / phases around the automatic choice
mkdir -p gpurun_out
{
python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 12288 --phases 12,16,20 --dict 0 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-200
python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 8192,10240 --phases 16 --dict 0 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-200
python tools/sweep.py webgoogle --S 44 --swz 1 --wpb 8 --win 10240 --phases 16 --dict 0 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-200
python tools/sweep.py webgoogle --S 56 --swz 1 --wpb 6 --win 12288,14336 --phases 16 --dict 0 --iters 2000 --check 2>&1 | grep -v "^#" | cut -c1-200
for L in 2 6; do CVR_WIN_LOADERS=$L python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 12288 --phases 16 --dict 0 --iters 2000 2>&1 | grep -v "^#" | cut -c1-200 | sed "s/^/loaders $L: /"; done
for nt in 2; do python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 12288 --phases 16 --dict 0 --nt 2 --iters 2000 2>&1 | grep -v "^#" | cut -c1-200 | sed "s/^/stream ahead 3: /"; done
} 2>&1 | tee gpurun_out/r03_dict_off_sweep.log

mkdir -p gpurun_out/r3c
{
python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 8192 --phases 12 --depth 1,2 --nt 0,2
python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 8192 --phases 12 --fold 18
python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 8192 --phases 12 --fold 12
python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 8192 --phases 12 --fold 1
python tools/sweep.py webgoogle --S 48 --swz 1 --wpb 7 --win 0 --phases 12
python tools/sweep.py webgoogle --S 24 --swz 1 --wpb 14 --win 8192 --phases 12 --thr 768 --depth 1,2
python tools/sweep.py webgoogle --S 32 --swz 1 --wpb 11 --win 8192 --phases 12 --thr 1024
python tools/sweep.py webgoogle --S 28 --swz 1 --wpb 12 --win 8192 --phases 12,24 --thr 896
python tools/sweep.py webgoogle --S 24 --swz 1 --wpb 14 --win 6144 --phases 12,24 --thr 768
python tools/sweep.py webgoogle --S 16 --swz 1 --wpb 16 --win 8192 --phases 12 --thr 512
} 2>&1 | grep -v "^$" > gpurun_out/r3c/sweep1.log
cat gpurun_out/r3c/sweep1.log

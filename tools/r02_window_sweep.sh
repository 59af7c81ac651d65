#!/bin/bash
# round 2: workgroups of several consecutive chunks sharing one LDS window of x (web-Google shape)
out=gpurun_out/r02_wg_window_sweep.log
: > $out
python tools/sweep.py webgoogle --S 56 --swz 1 --nt 0 --wpb 1 --win 0 --check >> $out 2>&1
python tools/sweep.py webgoogle --S 40,44,48 --swz 1 --nt 0 --wpb 8 --win 0,8192,12288,14336 --check >> $out 2>&1
python tools/sweep.py webgoogle --S 20,24 --swz 1 --nt 0 --wpb 16 --win 0,8192,12288 --check >> $out 2>&1
python tools/sweep.py webgoogle --S 28,32 --swz 1 --nt 0 --wpb 12 --win 8192,12288 --check >> $out 2>&1
python tools/sweep.py webgoogle --S 56,64,80,96 --swz 1 --nt 0 --wpb 4 --win 8192,12288,16384 --check >> $out 2>&1
python tools/sweep.py webgoogle --S 16,24,32 --swz 1 --nt 0 --wpb 8 --win 4096,6144 --check >> $out 2>&1
python tools/sweep.py webgoogle --S 44 --swz 1 --nt 0 --wpb 8 --win 12288 --depth 2 --check >> $out 2>&1
python tools/sweep.py webgoogle --S 44 --swz 1 --nt 0 --wpb 8 --win 12288 --dict 0 --check >> $out 2>&1
cat $out

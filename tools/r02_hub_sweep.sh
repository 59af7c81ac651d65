#!/bin/bash
# round 2: hub table on R-MAT shapes (numpy-built matrices of tools/sweep.py; the device-built ones of bench.py differ in their random streams)
L=gpurun_out/r02_hub_table_rmat.log; : > $L
echo "# R-MAT-22 fp32: plain layout, hub table per 8-chunk workgroup sizes, automatic rule (tools/sweep.py, every row checked)" >> $L
python tools/sweep.py rmat22 --S 32 --swz 1 --wpb 1 --win 0 --phases 0 --hub 0 --panels 1 --iters 50 --check 2>&1 >> $L
python tools/sweep.py rmat22 --S 32 --swz 1 --wpb 8 --win 0 --phases 0 --hub 0,8192,16384,24576,32768 --panels 1 --iters 50 --check 2>&1 | grep -v "^#" >> $L
python tools/sweep.py rmat22 --S 0 --swz 1 --wpb 0 --win -1 --phases -1 --hub=-1 --iters 50 --check 2>&1 | grep -v "^#" >> $L
echo "# R-MAT-22 fp64: automatic rule (table off: 41 % share) and the table forced" >> $L
python tools/sweep.py rmat64_22 --S 0 --swz 1 --wpb 0 --win -1 --phases -1 --hub=-1 --iters 50 --check 2>&1 | grep -v "^#" >> $L
python tools/sweep.py rmat64_22 --S 32 --swz 1 --wpb 8 --win 0 --phases 0 --hub=15360 --panels 1 --iters 50 --check 2>&1 | grep -v "^#" >> $L
echo "# counters, R-MAT-22 fp32 with the automatic hub table (tools/pmc_quick.sh)" >> $L
bash tools/pmc_quick.sh r02q_rmat22_hub rmat22 --S 0 --swz 1 --wpb 0 --win -1 --phases -1 --hub=-1 >> $L 2>&1
cat $L | cut -c1-250

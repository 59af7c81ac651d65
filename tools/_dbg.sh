python tools/sweep.py webgoogle --S 48 --swz 1,5,9,13 --wpb 7 --win 8192 --phases 12 --check 2>&1 | grep -v "^#" | cut -c1-200
python tools/sweep.py webgoogle --S 48 --swz 1,5,9,13 --wpb 7 --win 8192 --phases 12 --check 2>&1 | grep -v "^#" | cut -c1-200

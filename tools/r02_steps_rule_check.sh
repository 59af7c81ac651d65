#!/bin/bash
# round 2: the plain layout's rule for S (pick_steps, fitted in round 1 on row shards of the web-Google shape) checked on other shapes:
# for every shape S = 0 (the rule) and S = 8 .. 64, one chunk per workgroup, no window / phases / hub table (tools/sweep.py)
L=gpurun_out/r02_steps_rule_check.log; : > $L
for m in "webgoogle --scale 0.125" "webgoogle --scale 0.25" "rmat18" "rmat20" "band2e5" "band1e6" "livejournal --scale 0.0625" "livejournal --scale 0.25"; do
  echo "## $m" >> $L
  timeout 300 python tools/sweep.py $m --S 0,8,12,16,20,24,28,32,40,48,56,64 --swz 1 --wpb 1 --win 0 --phases 0 --hub 0 --panels 1 --iters 200 2>&1 | grep -v "^#   S" | cut -c1-110 >> $L
done
python3 - <<PY
import re
rows = {}; cur = None
for l in open("$L"):
    if l.startswith("## "): cur = l[3:].strip(); rows[cur] = []
    elif l.strip() and not l.startswith("#"):
        f = l.split(); rows[cur].append((int(f[0]), float(f[8])))
with open("$L", "a") as o:
    o.write("## summary: rule's S (first line of a block is S = 0: the chunk count tells which S it took) against the best measured\n")
    for k, v in rows.items():
        if not v: continue
        rule_t = v[0][1]; best = min(v[1:], key=lambda p: p[1])
        o.write(f"{k:32s} rule {rule_t:8.2f} us   best S = {best[0]:3d}: {best[1]:8.2f} us   rule / best = {rule_t / best[1]:.3f}\n")
print(open("$L").read()[-1200:])
PY

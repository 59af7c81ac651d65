# XCD panels on the soc-LiveJournal1 shape: chunk length (a panel runs on 32 CUs, so the plain layout's rule must count its chunks against 32)
mkdir -p gpurun_out/r3l
for S in 16 24 32 48 64; do for XP in 16 8; do
  CVR_XCD_PANELS=$XP python bench.py --workload livejournal --steps 200 --warmup 20 --no-cpu-baseline --steps-per-chunk $S > gpurun_out/r3l/lj_S$S.json 2> gpurun_out/r3l/lj_S$S.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3l/lj_S$S.json").read().strip().splitlines()[-1])
print("S $S panels", d["config"]["col_panels"], "chunks", d["config"]["chunks_rank0"], "us/step %.1f" % (d["ms_per_step"] * 1e3), "frac %.3f" % d["roofline"]["frac"], "wrong", d["verdict_wrong_rows"])
PY
done; done 2>&1 | tee gpurun_out/r3l/summary.log

# column panels, one per XCD at a time (eight panels per launch): soc-LiveJournal1 shape; CVR_XCD_PANELS=0: the round-2 form (9 panels, each over the whole chip)
mkdir -p gpurun_out/r3j
for XP in 0 8 16 24 32; do
  CVR_XCD_PANELS=$XP python bench.py --workload livejournal --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3j/lj_xp$XP.json 2> gpurun_out/r3j/lj_xp$XP.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3j/lj_xp$XP.json").read().strip().splitlines()[-1])
print("CVR_XCD_PANELS=$XP panels", d["config"]["col_panels"], "us/step %.1f" % (d["ms_per_step"] * 1e3), "kernel_us %.1f" % d["roofline"]["kernel_us"], "frac %.3f" % d["roofline"]["frac"], "wrong", d["verdict_wrong_rows"], "pre_s", d["preprocess"]["create_and_preprocess_wall_s"])
PY
done 2>&1 | tee gpurun_out/r3j/summary.log

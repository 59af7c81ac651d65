# R-MAT-22 fp32: one image with a hub table (the rule's choice) against 8 / 16 column panels, with per-panel hub tables (each panel over the whole chip)
mkdir -p gpurun_out/r3p
for ARGS in "--col-panels -1" "--col-panels 8" "--col-panels 16"; do
  python bench.py --workload rmat22 --steps 100 --warmup 10 --no-cpu-baseline $ARGS > gpurun_out/r3p/o.json 2> gpurun_out/r3p/o.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r3p/o.json") if l.startswith("{")][-1])
print("$ARGS", "panels", d["config"]["col_panels"], "S", d["config"]["steps_per_chunk"], "us/step %.1f" % (d["ms_per_step"] * 1e3), "frac %.3f" % d["roofline"]["frac"], "wrong", d["verdict_wrong_rows"])
PY
done 2>&1 | tee gpurun_out/r3p/summary.log

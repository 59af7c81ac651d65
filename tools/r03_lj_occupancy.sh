# soc-LiveJournal1 shape, XCD panels: workgroups per CU (LDS: the staged row sums of a chunk) against the staged write-out
for CAP in 4096 768 512 384 256; do
  CVR_YSTAGE_CAP=$CAP python bench.py --workload livejournal --steps 100 --warmup 10 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err
  python - <<PY
import json
d = json.loads([l for l in open("/tmp/o.json") if l.startswith("{")][-1])
print("ystage cap $CAP lds", d["config"]["lds_bytes_per_workgroup"], "us/step %.1f" % (d["ms_per_step"] * 1e3), "wrong", d["verdict_wrong_rows"])
PY
done

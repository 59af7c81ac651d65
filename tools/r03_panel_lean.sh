# column panels (one per XCD at a time) through the column-phase kernel (every piece carries its row): CVR_PANEL_PHASES=2
mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_parity.py -x -q -k "phases or timed or panel" 2>&1 | tail -3
for PP in 1 2; do for S in 0 16 24 32; do
  CVR_PANEL_PHASES=$PP python bench.py --workload livejournal --steps 200 --warmup 20 --no-cpu-baseline --steps-per-chunk $S > gpurun_out/r3m/lj.json 2> gpurun_out/r3m/lj.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3m/lj.json").read().strip().splitlines()[-1])
print("panel phases $PP S $S ->", d["config"]["steps_per_chunk"], "panels", d["config"]["col_panels"], "chunks", d["config"]["chunks_rank0"], "lds", d["config"]["lds_bytes_per_workgroup"], "us/step %.1f" % (d["ms_per_step"] * 1e3), "frac %.3f" % d["roofline"]["frac"], "wrong", d["verdict_wrong_rows"], "image MB", d["image_bytes"] // 1000000)
PY
done; done 2>&1 | tee gpurun_out/r3m/summary.log

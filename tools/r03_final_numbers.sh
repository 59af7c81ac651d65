# round-3 figures of the other shapes and the amortisation reports (profiles/r03_*.json)
O=gpurun_out/r03_final; mkdir -p $O
for W in livejournal rmat22 rmat24 banded3.5e6; do
  python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/bench_$W.json") if l.startswith("{")][-1])
print("$W", "us/step %.1f" % (d["ms_per_step"] * 1e3), "kernel_us %.1f" % d["roofline"]["kernel_us"], "frac %.3f" % d["roofline"]["frac"], "panels", d["config"]["col_panels"], "S", d["config"]["steps_per_chunk"], "wrong", d["verdict_wrong_rows"])
PY
done
export CVR_NO_TORCH_PRELOAD=1
for M in webgoogle livejournal; do python tests/compare_csr.py $M 200 > $O/cvr_vs_csr_$M.json 2> $O/cvr_vs_csr_$M.err; python - <<PY
import json
d = json.load(open("$O/cvr_vs_csr_$M.json"))
print("$M", "cvr %.1f us" % d["cvr"]["spmv_us"], "T_pre %.0f us" % d["cvr"]["preprocess_us"]["total"], {k: (round(v["spmv_us"], 1), round(v["I_pre_iterations"], 1) if v.get("I_pre_iterations") else None) for k, v in d["baselines"].items()})
PY
done

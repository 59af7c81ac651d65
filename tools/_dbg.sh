export PYTHONPATH=.
python tools/shard_kernel_probe.py > gpurun_out/r02_shard_kernels.log 2>&1; cat gpurun_out/r02_shard_kernels.log | cut -c1-420
python tests/compare_csr.py webgoogle 300 > gpurun_out/r02_cvr_vs_csr_webgoogle.json 2>&1
python tests/compare_csr.py livejournal 50 > gpurun_out/r02_cvr_vs_csr_livejournal.json 2>&1
python - <<PY
import json
for f in ("r02_cvr_vs_csr_webgoogle","r02_cvr_vs_csr_livejournal"):
    d=json.load(open(f"gpurun_out/{f}.json")); print(f, {k:round(v) for k,v in d["cvr"]["preprocess_us"].items()}, round(d["cvr"]["spmv_us"],2))
    for k,v in d["baselines"].items(): print("   ",k, round(v["spmv_us"],1), "own pre", round(v["own_preprocess_us"]), "I_pre", round(v["I_pre_iterations"],1), "with h2d", round(v["I_pre_iterations_with_h2d"],1), v["result_ok"])
PY

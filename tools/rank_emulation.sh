#!/bin/bash
# tools/rank_emulation.sh WORKLOAD [N] -- the per-rank regimes of an N-GPU configuration (BASELINE.json configs[3], [4]) on ONE GPU:
# for r = 0 .. N-1, bench.py builds only rank r's row shard (x replicated, as on N GPUs), times its SpMV and checks every row on
# the device.  The slowest rank is what the N-GPU job's compute-only step would take; the exchange is not part of it.
# Output: gpurun_out/rank_emulation_WORKLOAD.json (copy to profiles/).
W=${1:-rmat26}; N=${2:-8}
OUT=gpurun_out/rank_emulation_$W; mkdir -p $OUT
for r in $(seq 0 $((N-1))); do
  python bench.py --workload $W --emulate-rank $r/$N --steps 100 --warmup 10 --no-cpu-baseline > $OUT/rank$r.json 2> $OUT/rank$r.err
done
python - <<PY
import json, glob
ranks = []
for r in range($N):
    d = json.loads([l for l in open("$OUT/rank%d.json" % r) if l.startswith("{")][-1])
    c, ro = d["config"], d["roofline"]
    ranks.append({"rank": r, "rows": c["rank_rows"], "nnz": c["rank_nnz"], "kernel_us": ro["kernel_us"], "us_per_step": d["ms_per_step"] * 1e3,
                  "frac_of_8TBs": ro["frac"], "gflops": d["value"], "col_panels": c["col_panels"], "steps_per_chunk": c["steps_per_chunk"],
                  "chunks": c["chunks_rank0"], "wrong_rows": d["verdict_wrong_rows"], "image_bytes": d["image_bytes"]})
slow = max(x["us_per_step"] for x in ranks)
nnz = sum(x["nnz"] for x in ranks)
out = {"workload": "$W", "ranks_emulated": $N, "dtype": d["dtype"], "per_rank": ranks,
       "predicted_compute_only_us_per_step_on_%d_gpus" % $N: slow, "predicted_compute_only_gflops": 2.0 * nnz / slow / 1e3,
       "nnz_total": nnz, "all_rows_checked_wrong": sum(x["wrong_rows"] for x in ranks),
       "note": "every rank's shard built and timed alone on one MI355X with the full replicated x; the y all-gather of the real job is not included"}
json.dump(out, open("$OUT.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

#!/bin/bash
# the benchmark fluid at the sizes a brick of lj1m owns on 2, 4 and 8 GPUs (the scaling model's pass times), single domain, fused steps
out=gpurun_out; mkdir -p $out
for k in 79 63 50; do
  timeout 300 python bench.py --workload lj_side$k --no-cpu-baseline --no-secondary --traffic file --steps 3000 --warmup 500 > $out/r06_size_$k.json 2> $out/r06_size_$k.err
  python - <<PY
import json
try:
    d = json.load(open("$out/r06_size_$k.json")); r = d["roofline"]
    print("n_side $k", d["config"]["n_atoms"], "ms/step", round(d["ms_per_step"], 5), "k_forces us", round(r["avg_launch_ms"] * 1e3, 2), {a: round(b, 5) for a, b in r["stage_ms_per_step"].items() if b}, "blocks", d["engine"]["n_blocks"], d["engine"]["block_atoms"], d["engine"]["j_split"])
except Exception as e:
    print("n_side $k FAILED", e, open("$out/r06_size_$k.err").read()[-600:])
PY
done

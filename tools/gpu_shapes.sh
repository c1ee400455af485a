#!/bin/bash
# launch shapes of the pair pass at the sizes a brick of lj1m owns on 8, 4, 2 GPUs (and the whole box): ms/step of the fused single-domain step
out=gpurun_out; mkdir -p $out
for k in ${1:-50 63 79}; do
 for shape in "256 2" "256 4" "128 4" "128 8" "128 2" "64 8"; do
  set -- $shape
  timeout 300 python bench.py --workload lj_side$k --no-cpu-baseline --no-secondary --traffic file --steps 3000 --warmup 500 --block-atoms $1 --j-split $2 > $out/shape.json 2> $out/shape.err
  python - <<PY
import json
try:
    d = json.load(open("$out/shape.json")); r = d["roofline"]
    print("n_side $k shape $1x$2", d["config"]["n_atoms"], "ms/step", round(d["ms_per_step"], 5), "k_forces us", round(r["avg_launch_ms"] * 1e3, 2), {a: round(b, 5) for a, b in r["stage_ms_per_step"].items() if b}, "blocks", d["engine"]["n_blocks"], d["engine"]["block_atoms"], d["engine"]["j_split"], "fused", r["fused_step"], "slots", d["engine"]["n_list_slots"], "tile", d["engine"]["max_tile_atoms"])
except Exception as e:
    print("n_side $k shape $1x$2 FAILED", e, open("$out/shape.err").read()[-300:])
PY
 done
done

#!/bin/bash
# the reference's GPU memory-limit recipe on the GPU box (bench.py --workload memlimit); $1 = tag, $2 = --memlimit-max (0: until a size fails)
out=gpurun_out; mkdir -p $out; tag=${1:-r06}; max=${2:-0}
timeout ${3:-1500} python bench.py --workload memlimit --memlimit-max $max > $out/${tag}_memlimit.json 2> $out/${tag}_memlimit.err; echo "memlimit rc $?"
grep memlimit $out/${tag}_memlimit.err | tail -20
python - <<PY
import json
try:
    d = json.load(open("$out/${tag}_memlimit.json"))
    print("largest", d["value"], d["ms_per_step"], d["largest_that_ran"].get("hbm_in_use_gb"), "failed:", (d.get("smallest_that_failed") or {}).get("n_atoms"), (d.get("smallest_that_failed") or {}).get("error"))
    print("oracle", d["trials"][0].get("oracle_check"))
    for t in d["trials"]:
        print(t["n_atoms"], t["ok"], t.get("ms_per_step"), t.get("hbm_in_use_gb"), t.get("pairs_deviation_sigma"), t.get("net_force_over_abs_force"), t.get("block_atoms"), t.get("j_split"), t.get("max_tile_atoms"), t.get("bytes_per_atom_in_hbm"))
except Exception as e:
    print("FAILED", e)
PY

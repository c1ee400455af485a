#!/bin/bash
# one margin of the dual pair list swept over the fluids, alternating: tools/gpu_margin.sh [MOLLYHIP_OUTER_MARGIN_PM | MOLLYHIP_INNER_SKIN_PM] ["values in pm"]
out=gpurun_out; mkdir -p $out
VAR=${1:-MOLLYHIP_OUTER_MARGIN_PM}; for rep in 1 2; do for pm in ${2:-200 150 100 250}; do for wl in lj1m lj256k; do
    env $VAR=$pm timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary --traffic file --steps 6000 --warmup 1000 > $out/mg.json 2> $out/mg.err
    python - <<PY
import json
try:
    d = json.load(open("$out/mg.json")); r = d["roofline"]
    print("$VAR $pm $wl ms/step", round(d["ms_per_step"], 5), {a: round(b, 5) for a, b in r["stage_ms_per_step"].items() if b}, "per call", {a: round(b, 4) for a, b in r["stage_ms_per_call"].items()}, d["list_upkeep_in_profile_pass"])
except Exception as e:
    print("$VAR $pm $wl FAILED", e, open("$out/mg.err").read()[-300:])
PY
done; done; done

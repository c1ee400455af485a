#!/bin/bash
# a quick gate while working on the kernels: the parity tests that build, prune and walk lists (123 cases, two minutes) + the three bench workloads' stage timers
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_6mrr.py tests/test_gpu_cadence.py tests/test_gpu_edge_cases.py tests/test_gpu_triclinic.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -4
for wl in lj1m lj256k 6mrr_pme; do
    timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary --traffic file --steps 4000 --warmup 1000 > $out/g.json 2> $out/g.err
    python - <<PY
import json
try:
    d = json.load(open("$out/g.json")); r = d["roofline"]
    print("$wl ms/step", round(d["ms_per_step"], 5), "build per call ms", round(r["stage_ms_per_call"]["build_kernel"], 4), "prune per call", round(r["stage_ms_per_call"]["list_filter"], 4), {a: round(b, 5) for a, b in r["stage_ms_per_step"].items() if b})
except Exception as e:
    print("$wl FAILED", e, open("$out/g.err").read()[-300:])
PY
done

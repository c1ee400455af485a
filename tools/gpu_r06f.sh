#!/bin/bash
out=gpurun_out; mkdir -p $out
python tools/micro/tri_xl_check.py 35 1 0.00457 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_domain.py -q -x --timeout 800 -p no:cacheprovider -k "own_waits or read_late" 2>&1 | tail -5
for v in -1 8 2; do
  for wl in 6mrr_pme lj256k; do
    MOLLYHIP_TRANSPOSED_FROM_JS=$v timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 4000 --warmup 1000 > $out/tr.json 2> $out/tr.err
    python - <<PY
import json
try:
    d = json.load(open("$out/tr.json")); r = d["roofline"]
    print("transposed_from_js $v $wl ms/step", round(d["ms_per_step"], 5), "build per call ms", round(r["stage_ms_per_call"]["build_kernel"], 4), {a: round(b, 5) for a, b in r["stage_ms_per_step"].items() if b})
except Exception as e:
    print("transposed_from_js $v $wl FAILED", e, open("$out/tr.err").read()[-300:])
PY
  done
done

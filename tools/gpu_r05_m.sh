#!/bin/bash
# round 5, call m: the fused step (pair pass + integrator in one launch) — parity first (chunked continuation ==, cadence, CM removal, full-size parity), then A/B
out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py tests/test_gpu_implementations.py tests/test_gpu_energy_conservation.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree:MOLLYHIP_FUSE_STEP=0 tree tree:MOLLYHIP_FUSE_STEP=0 tree 2>&1 | cut -c1-330; done | tee $out/r05_m_fuse_ab.txt
timeout 900 python bench.py --no-cpu-baseline > $out/r05_m_default.json 2> $out/r05_m_default.err
python - <<PY
import json
d = json.load(open("$out/r05_m_default.json"))
print("default", d["ms_per_step"], d["roofline"]["kernel"][:30], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [(r["config"]["name"], round(r["ms_per_step"], 4), round(r["roofline"]["avg_launch_ms"], 5), round(r["roofline"]["frac"], 3)) for r in d.get("secondary", [])], d["roofline"]["stage_ms_per_step"])
PY

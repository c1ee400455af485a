#!/bin/bash
# round 4, first GPU call: the new parity / engine-loop tests, the oracle-vs-engine NVE trace, a 6mrr_pme baseline of this round's box
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pme.py tests/test_gpu_domain.py tests/test_gpu_bench_cli.py -m gpu -q --timeout 900 -p no:cacheprovider > $out/r04_a_tests.log 2>&1; echo "rc $?" >> $out/r04_a_tests.log
tail -30 $out/r04_a_tests.log
timeout 900 python tools/nve_drift.py --oracle 2000 > $out/r04_nve_oracle.json 2> $out/r04_nve_oracle.err; echo "nve rc $?"
timeout 600 python bench.py --workload 6mrr_pme --steps 2000 --warmup 1000 --profile-steps 400 > $out/r04_a_6mrr_pme.json 2> $out/r04_a_6mrr_pme.err; echo "bench rc $?"
python - <<PY
import json
d = json.load(open("$out/r04_a_6mrr_pme.json")); print(d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["stage_ms_per_step"], d.get("cpu_baseline", {}).get("value"), d.get("summary"))
n = json.load(open("$out/r04_nve_oracle.json"))[0]; print({k: v for k, v in n.items() if not k.startswith("E_")})
PY

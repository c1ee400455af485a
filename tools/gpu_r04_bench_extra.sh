#!/bin/bash
# round 4: the reference-shaped benchmark rows at the head (benchmark/benchmark_gpu_tiles.jl, benchmark/protein.jl)
out=gpurun_out; mkdir -p $out
timeout 900 python bench.py --workload argon4096 > $out/r04_bench_argon4096.json 2> $out/r04_bench_argon4096.err; echo "argon4096 rc $?"
timeout 900 python bench.py --workload 6mrr_rf32 > $out/r04_bench_6mrr_rf32.json 2> $out/r04_bench_6mrr_rf32.err; echo "6mrr_rf32 rc $?"
python - <<'PY'
import json
for n in ("argon4096", "6mrr_rf32"):
    try:
        d = json.loads(open(f"gpurun_out/r04_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d.get("metric"), d.get("value"), d.get("ms_per_step"), json.dumps(d.get("summary", d.get("config", {})))[:300])
    except Exception as e:
        print(n, "FAILED", e)
PY

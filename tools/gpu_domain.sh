#!/bin/bash
# the multi-process domain tests on the one GPU (engine loop with IPC peer stores), then the N = 1 box through the same loop against mhip_vv_run
out=gpurun_out; mkdir -p $out; tag=${1:-dom}
timeout 1500 python -m pytest tests/test_gpu_domain.py -x -q --timeout 600 -p no:cacheprovider > $out/${tag}_domain.log 2>&1; echo "rc $?" >> $out/${tag}_domain.log
tail -12 $out/${tag}_domain.log
timeout 600 python bench.py --steps 2000 --warmup 300 --no-cpu-baseline --no-secondary > $out/${tag}_vvrun.json 2> $out/${tag}_vvrun.err
MOLLYHIP_FORCE_DOMAIN=1 timeout 600 python bench.py --steps 2000 --warmup 300 --no-cpu-baseline > $out/${tag}_domain1.json 2> $out/${tag}_domain1.err
MOLLYHIP_FORCE_DOMAIN=1 MOLLYHIP_ENGINE_LOOP=0 timeout 600 python bench.py --steps 2000 --warmup 300 --no-cpu-baseline > $out/${tag}_domain1_py.json 2> $out/${tag}_domain1_py.err
TAG=$tag python - <<'PY'
import json
for f in ("vvrun","domain1","domain1_py"):
    try:
        import sys, os
        d=json.load(open(f"gpurun_out/{os.environ.get('TAG','dom1')}_{f}.json"))
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f, d["ms_per_step"], d["config"].get("parallelism"))
PY

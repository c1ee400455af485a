#!/bin/bash
# the round's gate on the GPU box, as the driver runs it: every -m gpu test, smoke(), bench.py in the driver's form and in its default form
out=gpurun_out; mkdir -p $out; tag=${1:-r05}
timeout 3000 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/${tag}_gputest.log 2>&1; echo "rc $?" >> $out/${tag}_gputest.log
tail -4 $out/${tag}_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1; tail -2 $out/${tag}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_20_5.json 2> $out/${tag}_bench_20_5.err; echo "bench 20/5 rc $?"
timeout 900 python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err; echo "bench default rc $?"
python - <<PY
import json
for f in ("${tag}_bench_20_5", "${tag}_bench_default"):
    try:
        d = json.load(open("$out/" + f + ".json"))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("window_ms_per_step"), [(r["config"]["name"], round(r["ms_per_step"], 4), round(r["value"], 1)) for r in d.get("secondary", [])], d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("one_core", {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY

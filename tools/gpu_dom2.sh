#!/bin/bash
# two ranks sharing the one GPU (gloo for the setup collectives): the engine loop with peer stores against the host loop with all_to_all
out=gpurun_out; mkdir -p $out
for loop in 1 0; do
  MOLLYHIP_ENGINE_LOOP=$loop MOLLYHIP_DIST_BACKEND=gloo MOLLYHIP_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2950$loop \
     bench.py --gpus 2 --workload ${1:-lj1m} --steps 600 --warmup 100 --equil 600 > $out/dom2_loop$loop.json 2> $out/dom2_loop$loop.err
  python - <<PY
import json
try:
    d = json.load(open("$out/dom2_loop$loop.json"))
    print("engine_loop=$loop", round(d["ms_per_step"], 4), d["config"]["parallelism"][:160], {k: round(v, 4) for k, v in d["roofline"]["stage_ms_per_step"].items() if v})
except Exception as e:
    print("engine_loop=$loop FAILED", e, open("$out/dom2_loop$loop.err").read()[-800:])
PY
done

#!/bin/bash
# round 5, call b: the re-plan tests with atoms that really migrate, benchmark-size bricks, N = 1 through the domain loop after the double-prune fix,
# the default record after the scaffolding strip, integrator block counts at 256k atoms
out=gpurun_out; mkdir -p $out
export MOLLYHIP_XFER_TIMEOUT_MS=8000
timeout 1500 python -m pytest tests/test_gpu_domain.py -q -k "device_replan or benchmark_size" --timeout 900 -p no:cacheprovider > $out/r05_b_devreplan.log 2>&1; echo "device_replan + benchmark-size tests rc $?"; grep -v "socket.cpp\|amdgpu.ids\|Gloo" $out/r05_b_devreplan.log | tail -40
for fd in 0 1; do
  if [ $fd = 1 ]; then export MOLLYHIP_FORCE_DOMAIN=1; else unset MOLLYHIP_FORCE_DOMAIN; fi
  timeout 900 python bench.py --workload lj1m --steps 2000 --warmup 500 --no-cpu-baseline --no-secondary > $out/r05_b_lj1m_fd$fd.json 2> $out/r05_b_lj1m_fd$fd.err
  python -c "
import json; d=json.load(open('$out/r05_b_lj1m_fd$fd.json')); print('force_domain=$fd', round(d['ms_per_step'],4), d['config']['parallelism'][:200], {k: round(v, 4) for k, v in d['roofline']['stage_ms_per_step'].items() if v})" || tail -5 $out/r05_b_lj1m_fd$fd.err
done
unset MOLLYHIP_FORCE_DOMAIN
for vb in 0 512 1024; do
  MOLLYHIP_VV_BLOCKS=$vb timeout 600 python bench.py --workload lj256k --steps 2000 --warmup 500 --no-cpu-baseline --no-secondary > $out/r05_b_lj256k_vb$vb.json 2> $out/r05_b_lj256k_vb$vb.err
  python -c "
import json; d=json.load(open('$out/r05_b_lj256k_vb$vb.json')); print('lj256k vv_blocks=$vb', round(d['ms_per_step'],4), {k: round(v, 4) for k, v in d['roofline']['stage_ms_per_step'].items() if v})" || tail -5 $out/r05_b_lj256k_vb$vb.err
done
timeout 900 python bench.py --no-cpu-baseline > $out/r05_b_default.json 2> $out/r05_b_default.err
python - <<PY
import json
d = json.load(open("$out/r05_b_default.json"))
print("default", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [(r["config"]["name"], round(r["ms_per_step"], 4), round(r["roofline"]["avg_launch_ms"], 5)) for r in d.get("secondary", [])], d["roofline"]["stage_ms_per_step"])
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/micro/replan_cost.py --device lj256k 2>&1 | grep -v -i "warning\|socket" | tail -4 | tee $out/r05_b_replan_cost_dev.txt

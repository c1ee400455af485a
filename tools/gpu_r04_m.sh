#!/bin/bash
# round 4, call m: the N = 1 box through the domain loop against mhip_vv_run, and where a re-plan's time goes
out=gpurun_out; mkdir -p $out
timeout 600 python tools/micro/replan_cost.py 2>&1 | grep -v Warning | tail -12 | tee $out/r04_m_replan_cost.txt
for fd in 0 1; do
  if [ $fd = 1 ]; then export MOLLYHIP_FORCE_DOMAIN=1; else unset MOLLYHIP_FORCE_DOMAIN; fi
  timeout 900 python bench.py --workload lj1m --steps 2000 --warmup 500 --no-cpu-baseline --no-secondary > $out/r04_m_lj1m_fd$fd.json 2> $out/r04_m_lj1m_fd$fd.err
  python -c "
import json; d=json.load(open('$out/r04_m_lj1m_fd$fd.json')); print('force_domain=$fd', round(d['ms_per_step'],4), d['config']['parallelism'][:140])"
done

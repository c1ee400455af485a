#!/bin/bash
# round 5, call a: the re-plan inside the engine (first light), the two micro-benchmarks the round-4 review asked for, N = 1 through the domain loop
out=gpurun_out; mkdir -p $out
export MOLLYHIP_XFER_TIMEOUT_MS=8000
( cd tools/micro && ./grid_barrier > ../../$out/r05_grid_barrier.txt 2>&1; ./vv_records > ../../$out/r05_vv_records.txt 2>&1 )
cat $out/r05_grid_barrier.txt | head -30; cat $out/r05_vv_records.txt
timeout 900 python -m pytest tests/test_gpu_domain.py -x -q -k "device_replan" --timeout 600 -p no:cacheprovider > $out/r05_a_devreplan.log 2>&1; echo "device_replan tests rc $?"; tail -40 $out/r05_a_devreplan.log
timeout 1500 python -m pytest tests/test_gpu_domain.py tests/test_gpu_triclinic.py -q --timeout 900 -p no:cacheprovider > $out/r05_a_domain_all.log 2>&1; echo "domain + triclinic rc $?"; tail -30 $out/r05_a_domain_all.log
for fd in 0 1; do
  if [ $fd = 1 ]; then export MOLLYHIP_FORCE_DOMAIN=1; else unset MOLLYHIP_FORCE_DOMAIN; fi
  timeout 900 python bench.py --workload lj1m --steps 2000 --warmup 500 --no-cpu-baseline --no-secondary > $out/r05_a_lj1m_fd$fd.json 2> $out/r05_a_lj1m_fd$fd.err
  python -c "
import json; d=json.load(open('$out/r05_a_lj1m_fd$fd.json')); print('force_domain=$fd', round(d['ms_per_step'],4), d['config']['parallelism'][:200], {k: round(v, 4) for k, v in d['roofline']['stage_ms_per_step'].items() if v})" || tail -5 $out/r05_a_lj1m_fd$fd.err
done
unset MOLLYHIP_FORCE_DOMAIN
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/micro/replan_cost.py --device lj256k 2>&1 | grep -v -i warning | tail -6 | tee $out/r05_a_replan_cost_dev.txt

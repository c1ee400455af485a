#!/bin/bash
# round 5, call e: where the 2-brick run without a margin breaks (size sweep, both planners, with the margin), the batched integrator A/B
out=gpurun_out; mkdir -p $out
export MOLLYHIP_XFER_TIMEOUT_MS=6000
for args in "2 32 0.0 20" "2 48 0.0 20" "2 56 0.0 20" "2 64 0.0 9" "2 64 0.0 12" "2 64 0.0 20 0.0" "2 64 0.2 25" "2 64 0.05 60" "4 64 0.0 20"; do
  timeout 300 python tools/micro/brick_check.py $args 2>&1 | grep "^world" | cut -c1-400
done 2>&1 | tee $out/r05_e_brick_check.txt
for vb in 1 0; do
  for wl in lj256k lj1m; do
    MOLLYHIP_VV_BATCH=$vb timeout 600 python bench.py --workload $wl --steps 2000 --warmup 500 --no-cpu-baseline --no-secondary > $out/r05_e_${wl}_batch$vb.json 2> $out/r05_e_${wl}_batch$vb.err
    python -c "
import json; d=json.load(open('$out/r05_e_${wl}_batch$vb.json')); print('$wl vv_batch=$vb', round(d['ms_per_step'],4), {k: round(v, 4) for k, v in d['roofline']['stage_ms_per_step'].items() if v})" || tail -5 $out/r05_e_${wl}_batch$vb.err
  done
done
timeout 900 python -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3

#!/bin/bash
out=gpurun_out; mkdir -p $out
MOLLYHIP_BUILD_WALK=0 python tools/build_breakdown.py > $out/x4_build_breakdown_transposed.log 2>&1
MOLLYHIP_BUILD_WALK=1 python tools/build_breakdown.py > $out/x4_build_breakdown_walk.log 2>&1
python tools/prune_breakdown.py > $out/x4_prune_breakdown.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/x4_gputest.log 2>&1; echo "rc $?" >> $out/x4_gputest.log
tail -3 $out/x4_gputest.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 1000 --warmup 300 --no-cpu-baseline > $out/x4_$name.json 2> $out/x4_$name.err; }
run walk100 MOLLYHIP_INNER_SKIN_PM=100
run walk135 MOLLYHIP_INNER_SKIN_PM=135

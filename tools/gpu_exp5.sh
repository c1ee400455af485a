#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/x5_gputest.log 2>&1; echo "rc $?" >> $out/x5_gputest.log
tail -5 $out/x5_gputest.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 2000 --warmup 500 --no-cpu-baseline > $out/x5_$name.json 2> $out/x5_$name.err; }
run default
run skin120 MOLLYHIP_INNER_SKIN_PM=120
run skin100_transposed MOLLYHIP_BUILD_WALK=0
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/x5_default_20_5.json 2>/dev/null
timeout 300 python bench.py --workload 6mrr_pme --steps 2000 --warmup 300 --no-cpu-baseline > $out/x5_6mrr_pme.json 2>/dev/null
timeout 300 python bench.py --workload lj256k --steps 2000 --warmup 300 --no-cpu-baseline > $out/x5_lj256k.json 2>/dev/null

#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/x6_gputest.log 2>&1; echo "rc $?" >> $out/x6_gputest.log
tail -5 $out/x6_gputest.log
run6() { name=$1; shift; env "$@" timeout 300 python bench.py --workload 6mrr_pme --steps 2000 --warmup 300 --no-cpu-baseline > $out/x6_6mrr_$name.json 2> $out/x6_6mrr_$name.err; }
run6 base
run6 b64j8 MOLLYHIP_BLOCK_I=64 MOLLYHIP_J_SPLIT=8
run6 b128j8 MOLLYHIP_BLOCK_I=128 MOLLYHIP_J_SPLIT=8
run6 b128j4 MOLLYHIP_BLOCK_I=128 MOLLYHIP_J_SPLIT=4
run6 b256j4 MOLLYHIP_BLOCK_I=256 MOLLYHIP_J_SPLIT=4
run6 overlap MOLLYHIP_OVERLAP=1
timeout 600 python bench.py --steps 20 --warmup 5 > $out/x6_default_20_5.json 2> $out/x6_default_20_5.err
timeout 300 python bench.py --steps 2000 --warmup 500 --no-cpu-baseline > $out/x6_default_2000.json 2>/dev/null

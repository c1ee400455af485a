#!/bin/bash
out=gpurun_out; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 2000 --warmup 500 --no-cpu-baseline > $out/x2_$name.json 2> $out/x2_$name.err; }
run base200_nosoa MOLLYHIP_INNER_SKIN_PM=200 MOLLYHIP_NO_SOA=1
run soa200 MOLLYHIP_INNER_SKIN_PM=200
run soa100 MOLLYHIP_INNER_SKIN_PM=100
run soa100_pk MOLLYHIP_INNER_SKIN_PM=100 MOLLYHIP_PRUNE_KERNEL=1
run soa200_pk MOLLYHIP_INNER_SKIN_PM=200 MOLLYHIP_PRUNE_KERNEL=1
run soa75_pk MOLLYHIP_INNER_SKIN_PM=75 MOLLYHIP_PRUNE_KERNEL=1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/x2_gputest.log 2>&1; echo "rc $?" >> $out/x2_gputest.log
tail -3 $out/x2_gputest.log
MOLLYHIP_PRUNE_KERNEL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cadence.py tests/test_gpu_domain.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/x2_gputest_pk.log 2>&1; echo "rc $?" >> $out/x2_gputest_pk.log
tail -3 $out/x2_gputest_pk.log

#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_domain.py -m gpu -q --timeout 900 -p no:cacheprovider > $out/x7_domain.log 2>&1; echo "rc $?" >> $out/x7_domain.log
tail -4 $out/x7_domain.log
MOLLYHIP_FORCE_DOMAIN=1 timeout 300 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline > $out/x7_domain_n1.json 2> $out/x7_domain_n1.err
MOLLYHIP_FORCE_DOMAIN=1 MOLLYHIP_HALO_OVERLAP=0 timeout 300 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline > $out/x7_domain_n1_noov.json 2> $out/x7_domain_n1_noov.err
timeout 300 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline > $out/x7_single.json 2>/dev/null
tail -2 $out/x7_domain_n1.err

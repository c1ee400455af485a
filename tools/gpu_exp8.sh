#!/bin/bash
out=gpurun_out; mkdir -p $out
MOLLYHIP_FORCE_DOMAIN=1 timeout 300 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline > $out/x8_domain_n1.json 2> $out/x8_domain_n1.err
MOLLYHIP_FORCE_DOMAIN=1 MOLLYHIP_HALO_OVERLAP=0 timeout 300 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline > $out/x8_domain_n1_noov.json 2> $out/x8_domain_n1_noov.err
tail -2 $out/x8_domain_n1.err

#!/bin/bash
# round 5, call ad: the lane's own record and row count looked at BEHIND the staging, the first round's index loads issued with them (tree) against ab/lib_base.so
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_implementations.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 ab/lib_base.so tree ab/lib_base.so:MOLLYHIP_FUSE_STEP=0 tree:MOLLYHIP_FUSE_STEP=0 ab/lib_base.so tree 2>&1 | cut -c1-330; done | tee $out/r05_ad_own_late.txt
timeout 600 python tools/force_ab.py --workload 6mrr_pme --steps 2000 ab/lib_base.so tree ab/lib_base.so tree 2>&1 | cut -c1-330 | tee -a $out/r05_ad_own_late.txt
echo finished

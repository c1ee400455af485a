#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pme.py tests/test_gpu_6mrr.py tests/test_gpu_cadence.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -3
timeout 1500 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree tree:MOLLYHIP_GROUP_SPLIT=0 tree 2>&1 | tee $out/r04_h_ab_6mrr.txt
timeout 1500 python tools/force_ab.py --workload 6mrr_direct --steps 1500 tree tree:MOLLYHIP_GROUP_SPLIT=0 2>&1 | tee $out/r04_h_ab_6mrr_direct.txt

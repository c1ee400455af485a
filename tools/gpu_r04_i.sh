#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_parity.py tests/test_gpu_cadence.py -q --timeout 900 -p no:cacheprovider > $out/r04_i_parity.log 2>&1; echo "rc $?" >> $out/r04_i_parity.log
tail -5 $out/r04_i_parity.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree:MOLLYHIP_SMALL_SORT=0 tree tree:MOLLYHIP_SMALL_SORT=0 tree 2>&1 | tee $out/r04_i_ab_6mrr.txt

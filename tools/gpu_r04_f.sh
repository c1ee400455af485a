#!/bin/bash
# round 4, call f: group-split pair pass (forces_gs.hip) — parity of the 6mrr / charged tests, then A/B against MOLLYHIP_GROUP_SPLIT=0
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_pme.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_cadence.py tests/test_gpu_stochastic.py -q --timeout 900 -p no:cacheprovider > $out/r04_f_parity.log 2>&1; echo "rc $?" >> $out/r04_f_parity.log
tail -6 $out/r04_f_parity.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree:MOLLYHIP_GROUP_SPLIT=0 tree tree:MOLLYHIP_GROUP_SPLIT=0 tree > $out/r04_f_ab_6mrr.txt 2>&1; cat $out/r04_f_ab_6mrr.txt
timeout 900 python tools/force_ab.py --workload 6mrr_direct --steps 1500 tree:MOLLYHIP_GROUP_SPLIT=0 tree > $out/r04_f_ab_6mrr_direct.txt 2>&1; cat $out/r04_f_ab_6mrr_direct.txt

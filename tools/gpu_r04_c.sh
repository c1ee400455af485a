#!/bin/bash
# round 4, call c: rebalanced inner lists — parity, then 6mrr / lj A/B (dynamic), then where the 6mrr pair kernel's time goes (static passes)
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_6mrr.py tests/test_gpu_edge_cases.py tests/test_gpu_cadence.py tests/test_gpu_triclinic.py -q --timeout 900 -p no:cacheprovider > $out/r04_c_parity.log 2>&1; echo "rc $?" >> $out/r04_c_parity.log
tail -5 $out/r04_c_parity.log
timeout 600 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree:MOLLYHIP_REBALANCE=0 tree tree:MOLLYHIP_REBALANCE=0 tree > $out/r04_c_ab_6mrr.txt 2>&1; cat $out/r04_c_ab_6mrr.txt
timeout 600 python tools/force_ab.py --workload 6mrr_pme --steps 300 --static tree:MOLLYHIP_REBALANCE=0 tree ab/lib_exp6.so ab/lib_exp7.so ab/lib_exp7.so:MOLLYHIP_REBALANCE=0 > $out/r04_c_static_6mrr.txt 2>&1; cat $out/r04_c_static_6mrr.txt
timeout 900 python tools/force_ab.py --workload lj256k --steps 1500 tree:MOLLYHIP_REBALANCE=0 tree > $out/r04_c_ab_lj256k.txt 2>&1; cat $out/r04_c_ab_lj256k.txt
timeout 900 python tools/force_ab.py --workload lj1m --steps 800 tree:MOLLYHIP_REBALANCE=0 tree > $out/r04_c_ab_lj1m.txt 2>&1; cat $out/r04_c_ab_lj1m.txt

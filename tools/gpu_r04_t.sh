#!/bin/bash
# round 4, call t: the outer list adopted as the inner list when the prune has nothing to drop (6mrr) — parity, A/B
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_pme.py tests/test_gpu_cadence.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -6 | tee $out/r04_t_tests.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 tree:MOLLYHIP_ADOPT_OUTER=0 tree tree:MOLLYHIP_ADOPT_OUTER=0 tree 2>&1 | tee $out/r04_t_ab_6mrr.txt
timeout 900 python tools/force_ab.py --workload 6mrr_rf32 --steps 2000 tree:MOLLYHIP_ADOPT_OUTER=0 tree 2>&1 | tee $out/r04_t_ab_6mrr_rf32.txt

#!/bin/bash
# round 4: the group-split pass's (block, group) items handed to its workgroups in a serpentine over their row counts (k_gs_balance) — parity, A/B against the build before
out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_pme.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -4 | tee $out/r04_bal_tests.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 ab/lib_before.so tree ab/lib_before.so tree 2>&1 | tee $out/r04_bal_ab_6mrr.txt
timeout 900 python tools/force_ab.py --workload 6mrr_rf32 --steps 2000 ab/lib_before.so tree 2>&1 | tee $out/r04_bal_ab_6mrr_rf32.txt

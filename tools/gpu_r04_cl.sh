#!/bin/bash
# round 4: the bonded slot sums with their dependent fetches in batches of four (bonded_collect_lane) — parity, A/B; lj1m against the build before the k_build masks (box check)
out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_pme.py tests/test_gpu_stochastic.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -4 | tee $out/r04_cl_tests.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 ab/lib_before.so tree ab/lib_before.so tree 2>&1 | tee $out/r04_cl_ab_6mrr.txt
timeout 900 python tools/force_ab.py --workload 6mrr_rf32 --steps 2000 ab/lib_before.so tree 2>&1 | tee $out/r04_cl_ab_6mrr_rf32.txt
timeout 900 python tools/force_ab.py --workload lj1m --steps 1500 ab/lib_premask.so tree 2>&1 | tee $out/r04_cl_ab_lj1m.txt

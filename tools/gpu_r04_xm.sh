#!/bin/bash
# round 4: exception lists as 128-bit offset masks in k_build — parity, A/B against the build before
out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_parity.py tests/test_gpu_cadence.py tests/test_gpu_edge_cases.py tests/test_gpu_triclinic.py tests/test_gpu_pme.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -5 | tee $out/r04_xm_tests.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 ab/lib_before.so tree ab/lib_before.so tree 2>&1 | tee $out/r04_xm_ab_6mrr.txt
timeout 900 python tools/force_ab.py --workload 6mrr_rf32 --steps 2000 ab/lib_before.so tree 2>&1 | tee $out/r04_xm_ab_6mrr_rf32.txt

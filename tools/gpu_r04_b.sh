#!/bin/bash
# round 4, call b: parity of everything that runs the generic / two-partner pair loop, the fixed tests of call a, then 6mrr A/B (PK2 against the one-partner loop)
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_6mrr.py tests/test_gpu_pme.py tests/test_gpu_implementations.py tests/test_gpu_edge_cases.py tests/test_gpu_triclinic.py tests/test_gpu_stochastic.py "tests/test_gpu_domain.py::test_engine_loop_replans_match_host_loop" -q --timeout 900 -p no:cacheprovider > $out/r04_b_parity.log 2>&1; echo "rc $?" >> $out/r04_b_parity.log
tail -5 $out/r04_b_parity.log
for wl in 6mrr_pme 6mrr_direct; do
  timeout 600 python tools/force_ab.py --workload $wl --steps 1500 ab/lib_nopk2.so - ab/lib_nopk2.so - > $out/r04_b_ab_$wl.txt 2>&1
  cat $out/r04_b_ab_$wl.txt
done

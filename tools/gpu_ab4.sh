#!/bin/bash
# A/B on the GPU box for the charged (generic-loop) workloads: parity of everything that runs the generic pair loop, then 6mrr_pme stage times
out=gpurun_out; mkdir -p $out; tag=${1:-ab}; shift
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_6mrr.py tests/test_gpu_pme.py tests/test_gpu_implementations.py tests/test_gpu_edge_cases.py tests/test_gpu_triclinic.py -x -q --timeout 900 -p no:cacheprovider > $out/${tag}_parity.log 2>&1; echo "rc $?" >> $out/${tag}_parity.log
tail -3 $out/${tag}_parity.log
timeout 600 python tools/force_ab.py --workload 6mrr_pme "$@" > $out/${tag}_6mrr.txt 2>&1
cat $out/${tag}_6mrr.txt

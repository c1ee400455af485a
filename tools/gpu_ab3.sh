#!/bin/bash
# A/B on the GPU box: parity subset (pair sets, cadence, 6mrr), then tools/force_ab.py with the given specs on lj1m and optionally 6mrr_pme
out=gpurun_out; mkdir -p $out; tag=${1:-ab}; shift
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cadence.py tests/test_gpu_6mrr.py tests/test_gpu_triclinic.py -x -q --timeout 600 -p no:cacheprovider > $out/${tag}_parity.log 2>&1; echo "rc $?" >> $out/${tag}_parity.log
tail -3 $out/${tag}_parity.log
timeout 1500 python tools/force_ab.py "$@" > $out/${tag}_force_ab.txt 2>&1
cat $out/${tag}_force_ab.txt
timeout 600 python tools/force_ab.py --workload 6mrr_pme "$@" > $out/${tag}_6mrr.txt 2>&1
cat $out/${tag}_6mrr.txt

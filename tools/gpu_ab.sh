#!/bin/bash
# A/B on the GPU box: VALU issue micro-benchmark, quick parity of the in-tree build, stage times of the given libraries
out=gpurun_out; mkdir -p $out; tag=${1:-ab}; shift
./tools/micro/valu_rate > $out/${tag}_valu_rate.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cadence.py -x -q --timeout 600 -p no:cacheprovider > $out/${tag}_parity.log 2>&1; echo "rc $?" >> $out/${tag}_parity.log
tail -3 $out/${tag}_parity.log
timeout 1500 python tools/force_ab.py "$@" > $out/${tag}_force_ab.txt 2>&1
cat $out/${tag}_force_ab.txt
head -12 $out/${tag}_valu_rate.txt

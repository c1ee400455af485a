#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-ab}; shift
timeout 1500 python tools/force_ab.py --static --steps 200 "$@" > $out/${tag}_static.txt 2>&1
cat $out/${tag}_static.txt
timeout 1500 python tools/force_ab.py --steps 600 "$@" > $out/${tag}_dyn.txt 2>&1
cat $out/${tag}_dyn.txt

#!/bin/bash
out=gpurun_out; mkdir -p $out
BUILD_BREAKDOWN_CASE=6mrr timeout 900 python tools/build_breakdown.py 2>&1 | tee $out/r04_n_build_breakdown_6mrr.txt

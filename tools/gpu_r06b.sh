#!/bin/bash
out=gpurun_out; mkdir -p $out
python bench.py --workload memlimit --memlimit-start 64000000 > $out/r06b_memlimit_top.json 2> $out/r06b_memlimit_top.err; echo "memlimit rc $?"
grep memlimit $out/r06b_memlimit_top.err | tail
timeout 1500 python -m pytest tests/test_gpu_large.py -q -x --timeout 1200 -p no:cacheprovider > $out/r06b_large.log 2>&1; tail -5 $out/r06b_large.log
bash profiles/collect.sh lj1m r06_lj1m 200 > $out/r06b_collect.log 2>&1; tail -3 $out/r06b_collect.log | cut -c1-300
ls $out/prof_r06_lj1m

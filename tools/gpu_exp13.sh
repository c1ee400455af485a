#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 300 python bench.py --steps 2000 --warmup 500 --no-cpu-baseline > $out/x13_default_2000.json 2>/dev/null
python -c "
import json; d=json.load(open('$out/x13_default_2000.json')); print('lj1m', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['stage_ms_per_step'])"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/x13_gputest.log 2>&1; echo "rc $?" >> $out/x13_gputest.log
tail -4 $out/x13_gputest.log

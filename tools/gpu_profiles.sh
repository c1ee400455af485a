#!/bin/bash
# full gate at this commit + the round's rocprofv3 profiles (summaries land in gpurun_out/prof_<tag>/)
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/r02_gputest.log 2>&1; echo "rc $?" >> $out/r02_gputest.log
tail -4 $out/r02_gputest.log
bash profiles/collect.sh lj1m r02_lj1m 300 > $out/collect_lj1m.log 2>&1
bash profiles/collect.sh lj256k r02_lj256k 300 > $out/collect_lj256k.log 2>&1
bash profiles/collect.sh 6mrr_pme r02_6mrr_pme 400 > $out/collect_6mrr.log 2>&1
ls -la $out/prof_r02_lj1m

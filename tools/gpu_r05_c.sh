#!/bin/bash
# round 5, call c: what the benchmark-size brick tests die of (wait reports, re-plan tables), the fp32 whole-step bars with their slack recorded
out=gpurun_out; mkdir -p $out; rm -f $out/tolerance_slack.jsonl
export MOLLYHIP_XFER_TIMEOUT_MS=6000 MOLLYHIP_DEBUG=1
timeout 900 python -m pytest tests/test_gpu_domain.py -q -k "benchmark_size" --timeout 600 -p no:cacheprovider > $out/r05_c_big.log 2>&1; echo "benchmark-size tests rc $?"
grep -v "socket.cpp\|amdgpu.ids\|Gloo" $out/r05_c_big.log | grep "MollyHipError\|re-plan at step\|passed\|failed\|FAILED\|assert" | sort | uniq -c | sort -rn | head -60
unset MOLLYHIP_DEBUG
timeout 900 python -m pytest tests/test_gpu_pme.py -q -k "fp32" -s --timeout 600 -p no:cacheprovider 2>&1 | grep "slack\|passed\|failed" | tee $out/r05_c_slack.txt

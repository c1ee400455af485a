#!/bin/bash
# round 5, call j: the single-list regression tests, the benchmark-size brick tests, then the whole gate (every -m gpu test, smoke)
out=gpurun_out; mkdir -p $out
export MOLLYHIP_XFER_TIMEOUT_MS=8000
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "single_pair_list" --timeout 600 -p no:cacheprovider 2>&1 | tail -5 | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_domain.py -q -k "benchmark_size" --timeout 900 -p no:cacheprovider > $out/r05_j_big.log 2>&1; echo "benchmark-size rc $?"; grep -v "socket.cpp\|amdgpu.ids\|Gloo" $out/r05_j_big.log | grep "passed\|failed\|FAILED\|^E  " | head -12 | cut -c1-300
timeout 3000 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/r05_j_gputest.log 2>&1; echo "rc $?" >> $out/r05_j_gputest.log
grep -v "socket.cpp\|amdgpu.ids\|Gloo" $out/r05_j_gputest.log | grep "passed\|failed\|FAILED\|^rc\|Error" | tail -15 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

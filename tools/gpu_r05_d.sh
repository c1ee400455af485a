#!/bin/bash
# round 5, call d: the 2-brick benchmark-size case without a margin — host planner against device planner, and another block shape
out=gpurun_out; mkdir -p $out
export MOLLYHIP_XFER_TIMEOUT_MS=6000
run() { echo "== $*"; env "$@" timeout 600 python -m pytest tests/test_gpu_domain.py -q -k "benchmark_size and 2-0.0" --timeout 500 -p no:cacheprovider 2>&1 | grep -v "socket.cpp\|amdgpu.ids\|Gloo" | grep "re-plan at step\|passed\|failed\|^E  .*assert\|MollyHipError" | sort | uniq -c | head -12; }
run MOLLYHIP_DEVICE_REPLAN=0 MOLLYHIP_DEBUG=0
run MOLLYHIP_DEVICE_REPLAN=1 MOLLYHIP_DEBUG=1 MOLLYHIP_BLOCK_I=128 MOLLYHIP_J_SPLIT=4
run MOLLYHIP_DEVICE_REPLAN=1 MOLLYHIP_DEBUG=1 MOLLYHIP_NO_UNIFORM_LJ=1
run MOLLYHIP_DEVICE_REPLAN=1 MOLLYHIP_DEBUG=1 MOLLYHIP_HALO_OVERLAP=0 MOLLYHIP_NO_SCALED_ENTRIES=1

#!/bin/bash
# round 5, call f: step-0 forces of two bricks without a ghost margin against the single domain, at the sizes / shapes where the runs of call e went wrong
out=gpurun_out; mkdir -p $out
for args in "2 32 0.0" "2 48 0.0" "2 48 0.2" "2 48 0.0 MOLLYHIP_BLOCK_I=64 MOLLYHIP_J_SPLIT=16" "2 48 0.0 MOLLYHIP_BUILD_WALK=0" "2 48 0.0 MOLLYHIP_NO_UNIFORM_LJ=1" "2 64 0.0" "2 48 0.0 MOLLYHIP_EXACT_OUTER=1"; do
  echo "== $args"; timeout 300 python tools/micro/brick_forces.py $args 2>&1 | grep "^rank\|Error\|error" | cut -c1-500
done 2>&1 | tee $out/r05_f_brick_forces.txt
timeout 900 python -X faulthandler -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py -v -x --timeout 600 -p no:cacheprovider > $out/r05_f_cadence_parity.log 2>&1; echo "cadence + parity rc $?"
grep -v "socket.cpp\|amdgpu.ids" $out/r05_f_cadence_parity.log | grep -v "PASSED" | tail -40 | cut -c1-300

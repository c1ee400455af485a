#!/bin/bash
# round 5, call g: what aborts test_gpu_consistency_lattice_100_atoms (last launch by MOLLYHIP_TRACE), the single pair list against the dual list by block shape
out=gpurun_out; mkdir -p $out
MOLLYHIP_TRACE=1 AMD_LOG_LEVEL=1 timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -k "lattice_100" -x -s --timeout 200 -p no:cacheprovider > $out/r05_g_lattice.log 2>&1; echo "lattice rc $?"
grep -v "socket.cpp\|amdgpu.ids" $out/r05_g_lattice.log | grep -v "^  File\|Extension modules" | tail -25 | cut -c1-300
for n in 32 40 48; do timeout 300 python tools/micro/nondual_check.py $n 2>&1 | grep "^n_side\|FAILED" | cut -c1-400; done | tee $out/r05_g_nondual.txt

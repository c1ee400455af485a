#!/bin/bash
# round 5, call h: exception lists with 128- and 256-atom blocks (a charged fluid of 46 656 / 110 592 atoms) against the oracle; the single list again with the XL = false search
out=gpurun_out; mkdir -p $out
for n in 36 48; do timeout 600 python tools/micro/xl_check.py $n 2>&1 | grep "n_side\|^   \|FAILED" | cut -c1-300; done | tee $out/r05_h_xl.txt
for n in 40 48; do timeout 300 python tools/micro/nondual_check.py $n 2>&1 | grep "^n_side\|FAILED" | cut -c1-300; done | tee $out/r05_h_nondual.txt
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q --timeout 600 -p no:cacheprovider > $out/r05_h_parity.log 2>&1; echo "parity rc $?"; grep -v "^  File\|Extension modules\|socket.cpp" $out/r05_h_parity.log | tail -8 | cut -c1-300

#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 300 python tools/micro/xl_waves.py 36 2>&1 | grep "shape\|block\|all blocks\|Error" | cut -c1-400
timeout 300 python tools/micro/xl_waves.py 36 MOLLYHIP_BUILD_WALK=0 2>&1 | grep "shape\|all blocks\|Error" | cut -c1-400
timeout 300 python tools/micro/xl_waves.py 36 MOLLYHIP_J_SPLIT=2 2>&1 | grep "shape\|all blocks\|Error" | cut -c1-400

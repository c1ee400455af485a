#!/bin/bash
# round 5, call o: where the fused epilogue's time goes — timing-only builds with parts of it switched off (MHIP_STEP_EXP bits: 1 no v_cm poll, 2 no stores, 4 no block sums, 8 no loads)
out=gpurun_out; mkdir -p $out
timeout 900 python tools/force_ab.py --workload lj1m --steps 600 tree:MOLLYHIP_FUSE_STEP=0 tree ab/lib_sx1.so ab/lib_sx2.so ab/lib_sx5.so ab/lib_sx8.so ab/lib_sx15.so 2>&1 | cut -c1-330 | tee $out/r05_o_step_parts.txt
echo finished

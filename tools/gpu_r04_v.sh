#!/bin/bash
# round 4, call v: the two searches of k_build on 6mrr (walk over the cell stencil against the transposed search)
out=gpurun_out; mkdir -p $out
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 tree tree:MOLLYHIP_BUILD_WALK=0 tree tree:MOLLYHIP_BUILD_WALK=0 2>&1 | tee $out/r04_v_ab_6mrr.txt

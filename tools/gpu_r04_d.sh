#!/bin/bash
# round 4, call d: the reciprocal + bonded chain on a stream of its own beside the pair kernel (MOLLYHIP_OVERLAP=2), plain and CU-partitioned
out=gpurun_out; mkdir -p $out
R=MOLLYHIP_REBALANCE=0
timeout 1500 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree:$R tree:$R,MOLLYHIP_OVERLAP=2 tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_J_SPLIT=8 tree:$R,MOLLYHIP_J_SPLIT=8 \
   tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_SIDE_CUS=32 tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_SIDE_CUS=64 tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_SIDE_CUS=64,MOLLYHIP_J_SPLIT=8 \
   tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_SIDE_CUS=32,MOLLYHIP_SIDE_CU_STRIDE=8 tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_SIDE_CUS=64,MOLLYHIP_SIDE_CU_STRIDE=4 \
   tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_SIDE_CUS=64,MOLLYHIP_MAIN_MASKED=0 tree:$R,MOLLYHIP_OVERLAP=2,MOLLYHIP_SIDE_CUS=8,MOLLYHIP_MAIN_MASKED=0 tree:$R > $out/r04_d_overlap.txt 2>&1
cat $out/r04_d_overlap.txt

#!/bin/bash
# round 5, call z: the first round's tile indices asked for at the top of the kernel (ab/lib_early.so; it spills 44-60 bytes outside the loops) against the tree
out=gpurun_out; mkdir -p $out
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree ab/lib_early.so tree:MOLLYHIP_FUSE_STEP=0 ab/lib_early.so:MOLLYHIP_FUSE_STEP=0 tree ab/lib_early.so 2>&1 | cut -c1-330; done | tee $out/r05_z_early_ab.txt
echo finished

#!/bin/bash
# round 5, call ab: the first round's tile indices asked for at the top of the kernel, lanes past the tile's end fetching ONE address (ab/lib_early2.so) against the tree
# (both with the row's LDS reads pinned)
out=gpurun_out; mkdir -p $out
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree ab/lib_early2.so tree:MOLLYHIP_FUSE_STEP=0 ab/lib_early2.so:MOLLYHIP_FUSE_STEP=0 tree ab/lib_early2.so 2>&1 | cut -c1-330; done | tee $out/r05_ab_early2.txt
echo finished

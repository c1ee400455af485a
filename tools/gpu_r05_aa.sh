#!/bin/bash
# round 5, call aa: the row's twelve LDS reads pinned into one run (sched_group_barrier) — on the tree's kernel (lib_pin) and on the variant that asks for the first
# round's tile indices at the top of the kernel (lib_earlypin: 44-60 bytes of spills outside the loops)
out=gpurun_out; mkdir -p $out
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree ab/lib_pin.so ab/lib_earlypin.so tree:MOLLYHIP_FUSE_STEP=0 ab/lib_pin.so:MOLLYHIP_FUSE_STEP=0 ab/lib_earlypin.so:MOLLYHIP_FUSE_STEP=0 tree ab/lib_pin.so ab/lib_earlypin.so 2>&1 | cut -c1-330; done | tee $out/r05_aa_pin_ab.txt
echo finished

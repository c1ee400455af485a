#!/bin/bash
# round 5, call v: the fused epilogue at raised wave priority (s_setprio 3) — stamps, A/B
out=gpurun_out; mkdir -p $out
lib=$PWD/ab/lib_stamps.so
for wl in lj256k lj1m; do
echo "== $wl fused"
MOLLYHIP_DBG_TIMES=150 MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload $wl --steps 600 --equil 200 2>&1 | grep -E "mhip dbg" | cut -c1-420 | tail -2
done | tee $out/r05_v_step_stamps.txt
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree:MOLLYHIP_FUSE_STEP=0 tree tree:MOLLYHIP_FUSE_STEP=0 tree 2>&1 | cut -c1-330; done | tee $out/r05_v_fuse_ab.txt
echo finished

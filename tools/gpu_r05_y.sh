#!/bin/bash
# round 5, call y (throwaway build ab/lib_stg.so): stamps inside the STAGING of the pair pass — columns: "staging" = entry -> the block's scalars and the lane's own
# record are there; "row walk" = the tile rounds (index -> coordinates -> LDS); "reduce+store" = the barrier behind them
out=gpurun_out; mkdir -p $out
lib=$PWD/ab/lib_stg.so
for wl in lj256k lj1m; do for fs in 0 1; do
echo "== $wl MOLLYHIP_FUSE_STEP=$fs"
MOLLYHIP_FUSE_STEP=$fs MOLLYHIP_DBG_TIMES=150 MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload $wl --steps 600 --equil 200 2>&1 | grep -E "mhip dbg" | cut -c1-420 | tail -2
done; done | tee $out/r05_y_stage_stamps.txt
echo finished

#!/bin/bash
# round 5, call w (throwaway build ab/lib_epi.so): stamps INSIDE the fused epilogue — column "staging" = staging + walk + j-split reduction, "row walk" = records
# arrive + v_cm + arithmetic, "reduce+store" = block sums + record stores
out=gpurun_out; mkdir -p $out
lib=$PWD/ab/lib_epi.so
for wl in lj256k lj1m; do
echo "== $wl fused"
MOLLYHIP_DBG_TIMES=150 MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload $wl --steps 600 --equil 200 2>&1 | grep -E "mhip dbg" | cut -c1-420 | tail -2
done | tee $out/r05_w_epi_stamps.txt
echo finished

#!/bin/bash
# round 5, call p: wall-clock stamps of the pair pass with and without the integrator in its epilogue (ab/lib_stamps.so)
out=gpurun_out; mkdir -p $out
lib=$PWD/ab/lib_stamps.so
for wl in lj256k lj1m; do for fs in 0 1; do
echo "== $wl MOLLYHIP_FUSE_STEP=$fs"
MOLLYHIP_FUSE_STEP=$fs MOLLYHIP_DBG_TIMES=150 MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload $wl --steps 600 --equil 200 2>&1 | grep -E "mhip dbg|AB_RESULT" | cut -c1-420 | tail -4
done; done | tee $out/r05_p_step_stamps.txt
echo finished

#!/bin/bash
# round 5, call s|t: the fused step with its block sums by DPP rows, then with the v_cm words requested early and together — bit identity with the separate integrator, stamps, A/B
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -5 | cut -c1-300
lib=$PWD/ab/lib_stamps.so
for wl in lj256k lj1m; do
echo "== $wl fused"
MOLLYHIP_DBG_TIMES=150 MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload $wl --steps 600 --equil 200 2>&1 | grep -E "mhip dbg" | cut -c1-420 | tail -2
done | tee $out/r05_t_step_stamps.txt
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree:MOLLYHIP_FUSE_STEP=0 tree tree:MOLLYHIP_FUSE_STEP=0 tree 2>&1 | cut -c1-330; done | tee $out/r05_t_fuse_ab.txt
echo finished

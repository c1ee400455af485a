#!/bin/bash
# round 5, call x: (i) no barrier in front of the fused epilogue's block sums (tree), (ii) the list rows as non-temporal loads (lib_nt), (iii) the velocity's line
# touched at the start of the block (lib_t0), (iv) both (lib_ntt0); the separate integrator with and without nt rows as the reference points
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_cadence.py -q -x --timeout 900 -p no:cacheprovider -k "fused or chunked or full_size" 2>&1 | tail -2 | cut -c1-300
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree:MOLLYHIP_FUSE_STEP=0 ab/lib_nt.so:MOLLYHIP_FUSE_STEP=0 tree ab/lib_nt.so ab/lib_t0.so ab/lib_ntt0.so tree ab/lib_nt.so 2>&1 | cut -c1-330; done | tee $out/r05_x_nt_ab.txt
echo finished

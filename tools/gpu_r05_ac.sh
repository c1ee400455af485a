#!/bin/bash
# round 5, call ac: the first staging round's index loads bounded by the tile's capacity and issued unconditionally (tree) against the same library without that (ab/lib_base.so)
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 ab/lib_base.so tree ab/lib_base.so:MOLLYHIP_FUSE_STEP=0 tree:MOLLYHIP_FUSE_STEP=0 ab/lib_base.so tree 2>&1 | cut -c1-330; done | tee $out/r05_ac_early3.txt
echo finished

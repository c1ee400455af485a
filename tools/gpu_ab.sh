#!/bin/bash
# tools/gpu_ab.sh "<workloads>" <variant> [<variant> ...]: alternating A/B runs of library builds / environments on one GPU box (tools/force_ab.py per workload).
# A variant is the in-tree build ("tree"), a library built by tools/build_variant.sh ("ab/lib_x.so"), either with NAME=VALUE pairs behind a colon
# ("tree:MOLLYHIP_FUSE_STEP=0").  Every experiment of profiles/r05_force_ab.txt §9-§10 was a call of this shape, e.g.
#     gpurun -- bash tools/gpu_ab.sh "lj256k lj1m" tree:MOLLYHIP_FUSE_STEP=0 tree tree:MOLLYHIP_FUSE_STEP=0 tree
# Per-wave time stamps: tools/build_variant.sh stamps -DMHIP_STAMPS=1, then MOLLYHIP_DBG_TIMES=150 MOLLYHIP_LIB_AB=$PWD/ab/lib_stamps.so python tools/force_ab.py --child …
out=gpurun_out; mkdir -p $out; wls=$1; shift
for wl in $wls; do timeout 1200 python tools/force_ab.py --workload $wl --steps ${AB_STEPS:-1000} "$@" 2>&1 | cut -c1-330; done | tee $out/ab_$(date +%H%M%S).txt

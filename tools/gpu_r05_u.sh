#!/bin/bash
# round 5, call u: what lies between the kernels of a fused step (rocprofv3 kernel trace of the 1M-atom fluid, fused and not)
out=$PWD/gpurun_out; mkdir -p $out; R=$PWD
for fs in 1 0; do
cd /tmp && export TMPDIR=/tmp && MOLLYHIP_FUSE_STEP=$fs timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_gap$fs -o g -- python $R/tools/force_ab.py --child --workload lj1m --steps 400 --equil 100 2>&1 | grep AB_RESULT | cut -c1-60; cd $R
f=$(find $out/prof_gap$fs -name "*kernel_trace.csv" | head -1); echo "== MOLLYHIP_FUSE_STEP=$fs"; python tools/kernel_gaps.py $f; rm -rf $out/prof_gap$fs
done | tee $out/r05_u_kernel_gaps.txt
echo finished

#!/bin/bash
# round 4: rocprofv3 timeline of a plain 6mrr_pme step at the head
out=gpurun_out; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -o tl -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f | tee $out/r04_timeline_6mrr_pme_head.txt; rm -rf gpurun_out/prof_tl

#!/bin/bash
# (record of an experiment: MOLLYHIP_GS_SHORT_FIRST=1 became the default, profiles/r04_force_ab.txt §16)
# round 4, call y: the spreading and bonded workgroups at the head of the fused launch's grid instead of its tail
out=gpurun_out; mkdir -p $out
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 tree tree:MOLLYHIP_GS_SHORT_FIRST=1 tree tree:MOLLYHIP_GS_SHORT_FIRST=1 2>&1 | tee $out/r04_y_ab_6mrr.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp && MOLLYHIP_GS_SHORT_FIRST=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_y -o y -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_y -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f | tee $out/r04_y_timeline.txt; rm -rf gpurun_out/prof_y

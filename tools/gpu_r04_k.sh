#!/bin/bash
# round 4, call k: the group-split pair pass on a low-priority stream beside the reciprocal + bonded chain (MOLLYHIP_OVERLAP=3)
out=gpurun_out; mkdir -p $out
MOLLYHIP_OVERLAP=3 timeout 900 python -m pytest tests/test_gpu_pme.py tests/test_gpu_6mrr.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -3
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree tree:MOLLYHIP_OVERLAP=3 tree tree:MOLLYHIP_OVERLAP=3 2>&1 | tee $out/r04_k_ab_6mrr.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp && MOLLYHIP_OVERLAP=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_k -o k -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_k -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f | tee $out/r04_k_timeline.txt; rm -rf gpurun_out/prof_k

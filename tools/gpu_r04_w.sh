#!/bin/bash
# (record of an experiment: the kernels it timed — k_pme_xy, k_pme_yx / k_pme_yz and their MOLLYHIP_PME_PLANES / MOLLYHIP_PME_DEBUG switches — were removed afterwards, profiles/r04_force_ab.txt §15)
# round 4, call w: the reciprocal space between z r2c and the potential mesh as two plane kernels (k_pme_yx, k_pme_yz) — parity, A/B, timeline
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pme.py tests/test_gpu_6mrr.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -6 | tee $out/r04_w_tests.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 tree:MOLLYHIP_PME_PLANES=0 tree tree:MOLLYHIP_PME_PLANES=0 tree 2>&1 | tee $out/r04_w_ab_6mrr.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_w -o w -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_w -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f | tee $out/r04_w_timeline.txt; rm -rf gpurun_out/prof_w

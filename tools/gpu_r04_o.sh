#!/bin/bash
# round 4, call o: interpolation + slot sums + integrator as one launch (step_tail.h) — parity, A/B, timeline
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pme.py tests/test_gpu_6mrr.py tests/test_gpu_cadence.py tests/test_gpu_stochastic.py -q --timeout 900 -p no:cacheprovider 2>&1 | tail -4
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree:MOLLYHIP_STEP_TAIL=0 tree tree:MOLLYHIP_STEP_TAIL=0 tree 2>&1 | tee $out/r04_o_ab_6mrr.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_o -o o -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_o -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f | tee $out/r04_o_timeline.txt; rm -rf gpurun_out/prof_o

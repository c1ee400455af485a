#!/bin/bash
# round 5, call n: the fused step with its records read ahead (MOLLYHIP_STEP_TOUCH rows before the end of a block's list): parity, then the sweep
out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_cadence.py tests/test_gpu_parity.py tests/test_gpu_implementations.py tests/test_gpu_energy_conservation.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300
for wl in lj256k lj1m; do timeout 900 python tools/force_ab.py --workload $wl --steps 1000 tree:MOLLYHIP_FUSE_STEP=0 tree:MOLLYHIP_STEP_TOUCH=0 tree:MOLLYHIP_STEP_TOUCH=2 tree tree:MOLLYHIP_STEP_TOUCH=8 tree:MOLLYHIP_STEP_TOUCH=1000 tree:MOLLYHIP_FUSE_STEP=0 tree 2>&1 | cut -c1-330; done | tee $out/r05_n_fuse_ab.txt
echo finished

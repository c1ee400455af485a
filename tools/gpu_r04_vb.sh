#!/bin/bash
# round 4: the integrator launch with more, shorter blocks once the pair pass before it has summed the Σ m v partials (MOLLYHIP_CM_IN_PAIR_PASS, MOLLYHIP_VV_BLOCKS)
out=gpurun_out; mkdir -p $out
timeout 600 python tools/force_ab.py --workload lj256k --steps 3000 tree:MOLLYHIP_CM_IN_PAIR_PASS=0 tree tree:MOLLYHIP_VV_BLOCKS=512 tree:MOLLYHIP_VV_BLOCKS=1024 2>&1 | tee $out/r04_vb_ab_lj256k.txt
timeout 600 python tools/force_ab.py --workload lj1m --steps 1500 tree:MOLLYHIP_CM_IN_PAIR_PASS=0 tree tree:MOLLYHIP_VV_BLOCKS=1024 2>&1 | tee $out/r04_vb_ab_lj1m.txt

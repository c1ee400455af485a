#!/bin/bash
# round 4, call s: inner skin of the 6mrr pair list (walk ∝ r³ against a rebuild — sort, search, prune, regroup — per inner-list lifetime)
out=gpurun_out; mkdir -p $out
timeout 1500 python tools/force_ab.py --workload 6mrr_pme --steps 2000 tree tree:MOLLYHIP_INNER_SKIN_PM=140 tree:MOLLYHIP_INNER_SKIN_PM=160 tree:MOLLYHIP_INNER_SKIN_PM=180 tree:MOLLYHIP_INNER_SKIN_PM=200 tree 2>&1 | tee $out/r04_s_ab_6mrr.txt
MOLLYHIP_DEBUG=1 timeout 300 python tools/force_ab.py --child --workload 6mrr_pme --steps 100 --equil 0 2>&1 | grep -i "skin\|margin" | head -8

#!/bin/bash
# round 4, call l: the two lanes of an atom level their entry counts at the end of a prune (two sub-lists per atom) — parity, then lj1m / lj256k A/B
out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cadence.py tests/test_gpu_edge_cases.py tests/test_gpu_energy_conservation.py tests/test_gpu_implementations.py -q --timeout 900 -p no:cacheprovider > $out/r04_l_parity.log 2>&1; echo "rc $?" >> $out/r04_l_parity.log
tail -5 $out/r04_l_parity.log
timeout 1500 python tools/force_ab.py --workload lj1m --steps 800 tree:MOLLYHIP_LEVEL_PAIRS=0 tree tree:MOLLYHIP_LEVEL_PAIRS=0 tree 2>&1 | tee $out/r04_l_ab_lj1m.txt
timeout 900 python tools/force_ab.py --workload lj256k --steps 1500 tree:MOLLYHIP_LEVEL_PAIRS=0 tree 2>&1 | tee $out/r04_l_ab_lj256k.txt

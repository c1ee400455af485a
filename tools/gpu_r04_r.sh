#!/bin/bash
# (record of an experiment: the proportional dealing was removed afterwards, profiles/r04_force_ab.txt §13)
# round 4, call r: lanes of a group dealt to the atoms in proportion to their entries (k_regroup) — parity, A/B, time stamps
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_pme.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -6 | tee $out/r04_r_tests.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 1500 tree:MOLLYHIP_GROUP_SPLIT=0 tree tree:MOLLYHIP_GROUP_SPLIT=0 tree 2>&1 | tee $out/r04_r_ab_6mrr.txt
MOLLYHIP_DBG_TIMES=100 MOLLYHIP_DBG_DUMP=$PWD/$out/gs_dump.bin MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 600 --equil 0 2>&1 | grep AB_RESULT | cut -c1-200
python tools/gs_times.py $out/gs_dump.bin | tee $out/r04_r_gs_times.txt; rm -f $out/gs_dump.bin

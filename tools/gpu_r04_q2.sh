#!/bin/bash
# round 4: per-wave time stamps of the group-split pair pass of 6mrr at the head (library with -DMHIP_EXP=11)
out=gpurun_out; mkdir -p $out
MOLLYHIP_DBG_TIMES=100 MOLLYHIP_DBG_DUMP=$PWD/$out/gs_dump.bin MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 600 --equil 0 2>&1 | grep AB_RESULT | cut -c1-120
python tools/gs_times.py $out/gs_dump.bin | tee $out/r04_q2_gs_times.txt; rm -f $out/gs_dump.bin

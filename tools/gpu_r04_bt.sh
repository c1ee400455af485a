#!/bin/bash
# round 4: per-wave time stamps of k_build on 6mrr (library with -DMHIP_EXP=11)
out=gpurun_out; mkdir -p $out
MOLLYHIP_DBG_TIMES=1000000 MOLLYHIP_DBG_DUMP_BUILD=$PWD/$out/build_dump.bin MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 200 --equil 0 2>&1 | grep AB_RESULT | cut -c1-160
python tools/build_times.py $out/build_dump.bin | tee $out/r04_bt_build_times.txt; rm -f $out/build_dump.bin

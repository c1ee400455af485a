#!/bin/bash
out=gpurun_out; mkdir -p $out
MOLLYHIP_DBG_TIMES=50 MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 --static > $out/r04_e_dbg_static.txt 2>&1; grep "mhip dbg\|AB_RESULT" $out/r04_e_dbg_static.txt | cut -c1-400
MOLLYHIP_DBG_TIMES=200 MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 600 --equil 0 > $out/r04_e_dbg_dyn.txt 2>&1; grep "mhip dbg\|AB_RESULT" $out/r04_e_dbg_dyn.txt | cut -c1-400

#!/bin/bash
# round 4, call q: per-wave time stamps of the group-split pair pass of 6mrr (library with -DMHIP_EXP=11)
out=gpurun_out; mkdir -p $out
for fuse in 1 0; do
  MOLLYHIP_GS_FUSE_SPREAD=$fuse MOLLYHIP_DBG_TIMES=100 MOLLYHIP_DBG_DUMP=$PWD/$out/gs_dump_$fuse.bin MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 600 --equil 0 2>&1 | grep AB_RESULT | cut -c1-300
  echo "== fused with the spreading and the bonded terms: $fuse" | tee -a $out/r04_q_gs_times.txt
  python tools/gs_times.py $out/gs_dump_$fuse.bin | tee -a $out/r04_q_gs_times.txt
  rm -f $out/gs_dump_$fuse.bin
done

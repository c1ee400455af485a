#!/bin/bash
# (record of experiments that were removed afterwards, profiles/r04_force_ab.txt §17; ab/lib_before.so / ab/lib_dbg.so are builds made for the comparison)
# round 4, call z: the first staging round's indices requested before the tile's size is known (forces_gs.hip) — parity, A/B against the build before, time stamps
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_6mrr.py tests/test_gpu_pme.py -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -4 | tee $out/r04_z_tests.log
timeout 900 python tools/force_ab.py --workload 6mrr_pme --steps 2000 ab/lib_before.so tree ab/lib_before.so tree 2>&1 | tee $out/r04_z_ab_6mrr.txt
MOLLYHIP_DBG_TIMES=100 MOLLYHIP_DBG_DUMP=$PWD/$out/gs_dump.bin MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 600 --equil 0 2>&1 | grep AB_RESULT | cut -c1-120
python tools/gs_times.py $out/gs_dump.bin | tee $out/r04_z_gs_times.txt; rm -f $out/gs_dump.bin

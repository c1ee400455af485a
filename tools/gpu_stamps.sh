#!/bin/bash
# Wall-clock stamps inside the block kernels of a small system (6mrr_pme): ab/lib_stamps.so = tools/build_variant.sh stamps -DMHIP_STAMPS=1 (built on the
# CPU side before the call).  Per-wave phases of the group-split pair pass (tools/gs_times.py) and of the search kernel (tools/build_times.py).
out=gpurun_out; mkdir -p $out; tag=${1:-r05}
lib=$PWD/ab/lib_stamps.so
MOLLYHIP_DBG_TIMES=100 MOLLYHIP_DBG_DUMP=$PWD/$out/gs_dump.bin MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 600 --equil 0 2>&1 | grep AB_RESULT | cut -c1-120
python tools/gs_times.py $out/gs_dump.bin | tee $out/${tag}_gs_times.txt; rm -f $out/gs_dump.bin
MOLLYHIP_DBG_TIMES=1000000 MOLLYHIP_DBG_DUMP=$PWD/$out/dump MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 200 --equil 0 2>&1 | grep AB_RESULT | cut -c1-160
python tools/build_times.py $out/dump.build | tee $out/${tag}_build_times.txt; rm -f $out/dump $out/dump.build $out/dump.regroup

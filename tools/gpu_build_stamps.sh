#!/bin/bash
# wall-clock stamps inside k_build for a workload (default lj1m): staging against search per block (ab/lib_stamps.so = tools/build_variant.sh stamps -DMHIP_STAMPS=1)
out=gpurun_out; mkdir -p $out; wl=${1:-lj1m}
lib=$PWD/ab/lib_stamps.so
MOLLYHIP_DBG_TIMES=1000000 MOLLYHIP_DBG_DUMP=$PWD/$out/dump MOLLYHIP_LIB_AB=$lib timeout 600 python tools/force_ab.py --child --workload $wl --steps 200 --equil ${2:-300} 2>&1 | grep AB_RESULT | cut -c1-200
python tools/build_times.py $out/dump.build ${3:-8} | tee $out/r06_build_times_$wl.txt; rm -f $out/dump $out/dump.build $out/dump.regroup

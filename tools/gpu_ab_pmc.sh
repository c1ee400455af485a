#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-pmc}
timeout 900 python tools/force_ab.py --static --steps 200 ab/libmollyhip_r02.so tree ab/lib_exp5.so > $out/${tag}_static.txt 2>&1
cat $out/${tag}_static.txt
MOLLYHIP_LIB_AB=$PWD/ab/libmollyhip_r02.so bash profiles/pmc_quick.sh ${tag}_r02 python $PWD/tools/force_ab.py --child --static --steps 100 --workload lj1m 2>&1 | tail -4
bash profiles/pmc_quick.sh ${tag}_tree python $PWD/tools/force_ab.py --child --static --steps 100 --workload lj1m 2>&1 | tail -4

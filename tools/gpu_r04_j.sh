#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 1500 python tools/force_ab.py --workload lj1m --steps 800 tree tree:MOLLYHIP_VV_THREADS=512 tree:MOLLYHIP_VV_THREADS=1024 tree:MOLLYHIP_VV_THREADS=1024,MOLLYHIP_VV_BLOCKS=256 tree:MOLLYHIP_VV_THREADS=512,MOLLYHIP_VV_BLOCKS=1024 tree 2>&1 | tee $out/r04_j_vv_lj1m.txt
timeout 900 python tools/force_ab.py --workload lj256k --steps 1500 tree tree:MOLLYHIP_VV_THREADS=512 tree:MOLLYHIP_VV_THREADS=1024 2>&1 | tee $out/r04_j_vv_lj256k.txt

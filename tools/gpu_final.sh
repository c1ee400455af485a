#!/bin/bash
# end-of-round evidence at the head: profiles of the three bench workloads, then the gate (tests, smoke, bench in both forms)
cd "$(dirname "$0")/.."; R=$PWD; out=$R/gpurun_out; mkdir -p $out
for wl in lj1m lj256k 6mrr_pme; do timeout 900 bash $R/profiles/collect.sh $wl r03_$wl 200 > $out/collect_$wl.log 2>&1; cd $R; done
bash tools/gpu_gate.sh r03
timeout 600 python tools/force_ab.py ab/libmollyhip_r02.so tree > $out/r03_vs_r02.txt 2>&1; cat $out/r03_vs_r02.txt | cut -c1-300

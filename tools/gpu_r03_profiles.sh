#!/bin/bash
# round-3 evidence at the head: rocprofv3 + PMC summaries of the three bench workloads, the NVE report, the two-rank loop comparison
cd "$(dirname "$0")/.."; R=$PWD; out=$R/gpurun_out; mkdir -p $out
for wl in lj1m lj256k 6mrr_pme; do timeout 900 bash $R/profiles/collect.sh $wl r03_$wl 200 > $out/collect_$wl.log 2>&1; cd $R; done
timeout 600 python tools/nve_drift.py --lj1m > $out/r03_nve_drift.json 2> $out/r03_nve_drift.err
timeout 900 bash tools/gpu_dom2.sh lj256k > $out/r03_dom2_lj256k.txt 2>&1
cat $out/r03_dom2_lj256k.txt; tail -c 600 $out/r03_nve_drift.json

#!/bin/bash
# The round's evidence at the head (tag = r05 …): rocprofv3 kernel statistics + PMC traffic of the three bench workloads (profiles/collect.sh), the
# rocprofv3 timeline of a plain 6mrr_pme step, the NVE report of the 1M-atom fluid, two ranks on the one GPU (engine loop against host loop).
# Summaries land in gpurun_out/; copy the ones to be judged into profiles/.
cd "$(dirname "$0")/.."; R=$PWD; out=$R/gpurun_out; mkdir -p $out; tag=${1:-r05}
for wl in lj1m lj256k 6mrr_pme; do timeout 900 bash $R/profiles/collect.sh $wl ${tag}_$wl 200 > $out/collect_$wl.log 2>&1; cd $R
  cp $out/prof_${tag}_$wl/${tag%%_*}_traffic_${wl}*.json $out/ 2>/dev/null      # per-kernel counter files with the library's ids (profiles/summarize.py): what bench.py's load_traffic reads
  python - <<PY
import json
try:
    d = json.load(open("$out/prof_${tag}_$wl/summary.json"))
    print("$wl", d.get("dominant_kernel"), {n: round(v["avg_us"], 2) for n, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ns"])[:8]}, d.get("hbm_bytes_per_force_launch"), d.get("lib_build_id"))
except Exception as e:
    print("$wl summary FAILED", e)
PY
done
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -o tl -- python $R/tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 > /dev/null 2>&1; cd $R
f=$(find $out/prof_tl -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f | tee $out/${tag}_timeline_6mrr_pme.txt; rm -rf $out/prof_tl
timeout 600 python tools/nve_drift.py --lj1m > $out/${tag}_nve_drift.json 2> $out/${tag}_nve_drift.err; tail -c 700 $out/${tag}_nve_drift.json
timeout 900 bash tools/gpu_dom2.sh lj256k > $out/${tag}_two_ranks_one_gpu_lj256k.txt 2>&1; cat $out/${tag}_two_ranks_one_gpu_lj256k.txt

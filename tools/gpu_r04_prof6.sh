#!/bin/bash
# round 4: rocprofv3 kernel statistics + PMC summary of 6mrr_pme at the head (profiles/collect.sh)
cd "$(dirname "$0")/.."; R=$PWD; out=$R/gpurun_out; mkdir -p $out
wl=6mrr_pme
timeout 900 bash $R/profiles/collect.sh $wl r04_$wl 200 > $out/collect_$wl.log 2>&1; cd $R
python - <<PY
import json
d = json.load(open("$out/prof_r04_$wl/summary.json"))
k = d.get("dominant_kernel", "k_forces")
t = {"workload": "$wl", "hbm_bytes_per_force_launch": d.get("hbm_bytes_per_force_launch"), "hbm_read_bytes_per_force_launch": d.get("hbm_read_bytes_per_force_launch"),
     "hbm_write_bytes_per_force_launch": d.get("hbm_write_bytes_per_force_launch"),
     "source": "round 4: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, profiles/collect.sh), FETCH_SIZE x2 gfx950 correction, KiB units; kernel " + k + " (non-pruning passes); profiles/r04_${wl}_summary.json"}
json.dump(t, open("$out/r04_traffic_$wl.json", "w"), indent=1)
print("$wl", k, {n: round(v["avg_us"], 2) for n, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ns"])[:12]}, t["hbm_bytes_per_force_launch"])
PY

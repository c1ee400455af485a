#!/bin/bash
# end-of-round record: smoke(), the gate (the round-1 failing test three times + the whole suite), the driver-form and the long bench
# lines for lj1m, the other workloads, the fp64 NVE drift report
out=gpurun_out; mkdir -p $out; tag=${1:-r02}
python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "rc $?" >> $out/${tag}_smoke.log; tail -2 $out/${tag}_smoke.log
bash tools/gpu_gate.sh $tag
for wl in lj256k 6mrr_pme 6mrr_direct 6mrr_rf64; do timeout 600 python bench.py --workload $wl --steps 2000 --warmup 300 > $out/${tag}_bench_$wl.json 2> $out/${tag}_bench_$wl.err; done
timeout 1200 python tools/nve_drift.py > $out/${tag}_nve_drift.json 2> $out/${tag}_nve_drift.err

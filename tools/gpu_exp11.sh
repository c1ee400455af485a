#!/bin/bash
out=gpurun_out; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 1000 --warmup 300 --no-cpu-baseline > $out/x11_$name.json 2> $out/x11_$name.err; }
run cur
(cd molly.jl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -fno-slp-vectorize -DMHIP_AB_NO_TRI_LOCAL -c forces_uniform.hip -o build/forces_uniform.o && make 2>&1 | tail -1) > $out/x11_rebuild.log 2>&1
run notri
run notri2
python - <<'PY'
import json
for n in ('cur','notri','notri2'):
    d=json.load(open(f'gpurun_out/x11_{n}.json')); print(n, d['ms_per_step'], d['roofline']['avg_launch_ms'])
PY

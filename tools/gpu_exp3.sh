#!/bin/bash
out=gpurun_out; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 1000 --warmup 300 --no-cpu-baseline > $out/x3_$name.json 2> $out/x3_$name.err; }
run gen200 MOLLYHIP_INNER_SKIN_PM=200 MOLLYHIP_NO_SOA=1
run gen200_pad12 MOLLYHIP_INNER_SKIN_PM=200 MOLLYHIP_NO_SOA=1 MOLLYHIP_LDS_PAD_KB=12
run soa100 MOLLYHIP_INNER_SKIN_PM=100
run soa135 MOLLYHIP_INNER_SKIN_PM=135
# SoA stride 3073 (36 KiB): rebuild the one translation unit
(cd molly.jl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -fno-slp-vectorize -DMHIP_SOA_STRIDE=3073 -c forces_uniform.hip -o build/forces_uniform.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -DMHIP_SOA_STRIDE=3073 -c engine.hip -o build/engine.o && make 2>&1 | tail -2) > $out/x3_rebuild.log 2>&1
run soa100_s3073 MOLLYHIP_INNER_SKIN_PM=100
run soa135_s3073 MOLLYHIP_INNER_SKIN_PM=135
python tools/build_breakdown.py > $out/x3_build_breakdown.log 2>&1

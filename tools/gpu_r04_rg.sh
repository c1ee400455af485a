#!/bin/bash
# round 4: wall-clock stamps inside k_regroup on 6mrr (library with -DMHIP_EXP=11)
out=gpurun_out; mkdir -p $out
MOLLYHIP_DBG_TIMES=1000000 MOLLYHIP_DBG_DUMP_REGROUP=$PWD/$out/rg_dump.bin MOLLYHIP_LIB_AB=$PWD/ab/lib_dbg.so timeout 600 python tools/force_ab.py --child --workload 6mrr_pme --steps 300 --equil 0 2>&1 | grep AB_RESULT | cut -c1-100
python - <<'PY' | tee gpurun_out/r04_rg_times.txt
import numpy as np
d = np.fromfile("gpurun_out/rg_dump.bin", dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 4] != 0].astype(np.int64)
t = (d[:, 0:5] - d[:, 0].min()) * 0.01
print(f"{len(d)} blocks; first entry -> last exit {t[:, 4].max():.1f} us; starts within {t[:, 0].max():.1f}")
for k, n in enumerate(("count (row loads)", "prefix + deal + rows", "scatter (row loads + LDS stores) + padding", "copy out")):
    a = t[:, k + 1] - t[:, k]
    print(f"  {n:44s} mean {a.mean():.2f} p50 {np.median(a):.2f} p90 {np.percentile(a, 90):.2f} max {a.max():.2f} us")
PY
rm -f gpurun_out/rg_dump.bin

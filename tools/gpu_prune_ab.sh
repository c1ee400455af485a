#!/bin/bash
# static timing of the FIRST pass (the prune pass over the outer list) for library variants: mhip_forces right after set_state+stale
python - "$@" <<'PY'
import os, sys, subprocess, json
root = os.getcwd()
code = r'''
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench, molly_loader, numpy as np, torch
m = molly_loader.load(); L = m.lib()
case, dtype, dt = bench.make_case("lj1m")
s = case.system(m, dtype); s.push_state(velocities=True); ctx = s.engine()
f = torch.empty((case.n, 3), dtype=torch.float32, device="cuda")
s._check(L.mhip_forces(ctx, 0, 0, f.data_ptr(), None, 1))
s._check(L.mhip_set_profiling(ctx, 1))
for k in range(8):
    s._check(L.mhip_forces(ctx, 10 * (k + 1), 0, f.data_ptr(), None, 1))   # MOLLYHIP_STRICT_CADENCE=1: every rebuild step re-prunes the outer list
st = s.stats()
print("PRUNE_US", 1e3 * st["prof_ms"][4] / max(st["prof_calls"][4], 1), st["prof_calls"][4], "build", 1e3 * st["prof_ms"][1] / max(st["prof_calls"][1], 1))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, MOLLYHIP_STRICT_CADENCE="1")
    if lib != "tree": env["MOLLYHIP_LIB_AB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(lib, [l for l in r.stdout.splitlines() if "PRUNE_US" in l] or r.stderr[-600:])
PY

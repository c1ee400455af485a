"""6mrr (fp32, PME) through mhip_langevin_run or mhip_vv_run: ms/step over 3000 steps (for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/micro/langevin_6mrr.py langevin)"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import molly_loader  # noqa: E402

m = molly_loader.load()
W = importlib.import_module("molly_jl_amd.workloads")
T = np.float32
case = W.protein_6mrr("ewald", T, pme=True)
s = case.system(m, T)
which = sys.argv[1] if len(sys.argv) > 1 else "langevin"
sim = m.Langevin(dt=0.0005, temperature=298.0, friction=1.0) if which == "langevin" else m.VelocityVerlet(dt=0.0005)
m.simulate(s, sim, 500, rng=1)
t = time.perf_counter()
m.simulate(s, sim, 3000, init_step=500, rng=1)
dt = time.perf_counter() - t
st = s.stats()
print(which, "ms/step", round(1e3 * dt / 3000, 5), "fused steps", st["n_fused_steps"], "outer builds", st["n_outer_builds"], "prunes", st["n_filter_passes"])

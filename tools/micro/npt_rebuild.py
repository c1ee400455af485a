"""what the first run behind a box change pays (6mrr, fp32, PME): wall time of 1-step and 30-step runs behind mhip_set_box against runs with nothing in between"""
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
sim = m.Langevin(dt=0.0005, temperature=298.0, friction=1.0)
m.simulate(s, sim, 200, rng=1)
L = m.lib()
step = 200
for tag, box_change in (("plain", False), ("behind set_box", True), ("plain", False), ("behind set_box", True)):
    for n in (1, 30):
        ts = []
        for it in range(12):
            if box_change:
                s.boundary = m.CubicBoundary(*s.boundary.side_lengths); s._push_box()
            s.push_state(velocities=True)
            st0 = s.stats()
            t = time.perf_counter()
            s._check(L.mhip_langevin_run(s._ctx, step, n, 0.0005, m.BOLTZMANN * 298.0, 1.0, 1, 1, 2)); step += n
            ts.append(time.perf_counter() - t)
            st1 = s.stats()
        print(f"{tag:16s} {n:3d} steps: {1e3 * np.mean(ts[2:]):7.3f} ms   rebuilds {st1['n_rebuilds'] - st0['n_rebuilds']} outer {st1['n_outer_builds'] - st0['n_outer_builds']} prunes {st1['n_filter_passes'] - st0['n_filter_passes']} last_rebuild_ms {st1['last_rebuild_ms']:.3f}")

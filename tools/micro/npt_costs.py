"""Where a Monte-Carlo barostat trial's time goes on 6mrr (fp32, PME): wall time of each call of one trial, averaged (one MI355X).   python tools/micro/npt_costs.py"""
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
acc = {}


def timed(name, f):
    t = time.perf_counter(); r = f(); acc.setdefault(name, []).append(time.perf_counter() - t); return r


for it in range(40):
    timed("run 30 steps", lambda: m.simulate(s, sim, 30, init_step=200 + 30 * it, rng=1))
    timed("PE (lists alive)", lambda: m.potential_energy(s))
    old_x, old_b = s.coords.copy(), s.boundary
    timed("scale_coords (host)", lambda: m.scale_coords(s, np.diag([1.001] * 3)))
    timed("set_box alone", lambda: s._push_box())
    timed("PE (new box: set_state + rebuild + three energies)", lambda: m.potential_energy(s))
    s.coords[:] = old_x; s.boundary = old_b
    timed("set_box back", lambda: s._push_box())
    timed("push_state back", lambda: s.push_state(velocities=True))
for k, v in acc.items():
    v = np.array(v[5:]) * 1e3
    print(f"{k:60s} {v.mean():8.3f} ms  (min {v.min():.3f})")

# fixed cost of a chunk: the same 30 steps (a) back to back, nothing in between, (b) behind a set_box to the same box, (c) as one 300-step call
import ctypes as C
acc.clear()
step = 5000
for it in range(30):
    timed("30 steps, nothing in between", lambda: m.simulate(s, sim, 30, init_step=step, rng=1)); step += 30
for it in range(30):
    s.boundary = m.CubicBoundary(*s.boundary.side_lengths)
    s._push_box()
    timed("30 steps behind set_box", lambda: m.simulate(s, sim, 30, init_step=step, rng=1)); step += 30
for it in range(10):
    timed("300 steps", lambda: m.simulate(s, sim, 300, init_step=step, rng=1)); step += 300
s.push_state(velocities=True)
for it in range(30):
    t = time.perf_counter()
    s._check(L.mhip_langevin_run(s._ctx, step, 30, 0.0005, m.BOLTZMANN * 298.0, 1.0, 1, 1, 2)); step += 30
    acc.setdefault("mhip_langevin_run(30) alone", []).append(time.perf_counter() - t)
for k, v in acc.items():
    v = np.array(v[3:]) * 1e3
    print(f"{k:60s} {v.mean():8.3f} ms  (min {v.min():.3f})")
print(s.stats()["n_rebuilds"], s.stats()["n_outer_builds"], s.stats()["n_box_changes"])

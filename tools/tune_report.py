"""mhip_optimize_launch_config on the bench workloads: python tools/tune_report.py [workload ...]  (needs an MI355X)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader  # noqa: E402
import bench  # noqa: E402

m = molly_loader.load()
for wl in (sys.argv[1:] or ["lj1m", "lj256k", "6mrr_pme"]):
    case, dtype, dt = bench.make_case(wl)
    s = case.system(m, dtype)
    m.simulate(s, m.VelocityVerlet(dt=dt), 300 if wl.startswith("lj") else 50)      # off the lattice
    st = s.stats()
    trials = m.optimize_launch_config(s, n_passes=20)
    print(wl, "automatic", (st["block_atoms"], st["j_split"]), "trials", [(a, b, round(c, 1)) for a, b, c in trials], flush=True)
    s.close()

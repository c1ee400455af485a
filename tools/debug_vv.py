import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader; m = molly_loader.load()
from tests import systems as S
case = S.charged_fluid(10, dict(kind="rf", rc=1.0), dtype=np.float64, with_exceptions=True, stable=True)
for n in (29, 30, 31, 39, 40, 41):
    o = case.oracle(np.float64); o.vv_run(n, 0.0005, remove_cm_every=1)
    s = case.system(m, np.float64); m.simulate(s, m.VelocityVerlet(dt=0.0005), n)
    st = s.stats()
    d=np.abs(s.velocities - o.vel); print(n, "n bad v", int((d.max(axis=1)>1e-6).sum()), "max dx", np.abs(s.coords - o.coords).max(), "dv", np.abs(s.velocities - o.vel).max(), st["n_outer_builds"], st["n_filter_passes"], st["minimg_mode"], st["block_atoms"], st["j_split"])

import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import molly_loader
pkg = molly_loader.load()
from tests import systems as S
from tests import golden6mrr as G
KB = pkg.BOLTZMANN
def lj(fuse, remove_cm):
    os.environ["MOLLYHIP_FUSE_STEP"] = fuse
    case = S.lj_fluid(40, seed=2, dtype=np.float32)
    s = case.system(pkg, np.float32)
    sim = pkg.Langevin(dt=0.002, temperature=85.0, friction=1.0, remove_CM_motion=remove_cm)
    pkg.simulate(s, sim, 20, rng=31); pkg.simulate(s, sim, 60, init_step=20, rng=32)
    return s.coords.astype(np.float64), s.velocities.astype(np.float64), case.box
for rc in (1, 0):
    a = lj("1", rc); b = lj("0", rc)
    d = a[0] - b[0]; d -= np.round(d / a[2]) * a[2]
    print("lj langevin fused vs unfused, remove_cm", rc, "max |dx|", np.abs(d).max(), "(bar 2e-5)  max |dv|", np.abs(a[1] - b[1]).max(), "(bar 5e-3)")
os.environ.pop("MOLLYHIP_FUSE_STEP", None)
def mrr(fuse, andersen):
    os.environ["MOLLYHIP_FUSE_GATHER_VV"] = fuse
    case = G.case("ewald", np.float32, bonded=True, pme=True)
    s = case.system(pkg, np.float32)
    sim = pkg.Langevin(dt=0.0005, temperature=300.0, friction=1.0, remove_CM_motion=1, coupling=pkg.AndersenThermostat(300.0, 0.05) if andersen else None)
    pkg.simulate(s, sim, 25, rng=9); pkg.simulate(s, sim, 15, init_step=25, rng=10)
    return np.array(s.coords, dtype=np.float64), np.array(s.velocities, dtype=np.float64)
for an in (False, True):
    a = mrr("1", an); b = mrr("0", an)
    box = G.data()["box"]; d = a[0] - b[0]; d -= np.round(d / box) * box
    print("6mrr langevin fused vs unfused, andersen", an, "max |dx|", np.abs(d).max(), "(bar 4e-6)  max |dv|", np.abs(a[1] - b[1]).max(), "(bar 4e-3)")

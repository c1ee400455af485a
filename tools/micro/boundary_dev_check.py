import os, sys, copy, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import molly_loader
pkg = molly_loader.load()
from tests import systems as S
from tests import golden6mrr as G
from tests.test_gpu_boundary import on_box, make, everything
for rep in range(3):
    case = G.case("ewald", np.float32, pme=True)
    s = case.system(pkg, np.float32)
    sim = pkg.VelocityVerlet(dt=0.0005, remove_CM_motion=1)
    pkg.simulate(s, sim, 10)
    pkg.scale_coords(s, np.diag([1.004] * 3))
    full = dict(specific=True, general=True)
    f1, e1 = pkg.forces(s, **full).astype(np.float64), pkg.potential_energy(s, **full)
    case2 = on_box(case, np.float32, s.coords, s.boundary.side_lengths)
    s2 = case2.system(pkg, np.float32, velocities=s.velocities)
    f2, e2 = pkg.forces(s2, **full).astype(np.float64), pkg.potential_energy(s2, **full)
    fmax = np.linalg.norm(f2, axis=1).max()
    pkg.simulate(s, sim, 20, init_step=10); pkg.simulate(s2, sim, 20, init_step=10)
    print("6mrr scaled box: df/fmax", np.linalg.norm(f1 - f2, axis=1).max() / fmax, "(bar 2e-5) de/e", abs(e1 - e2) / abs(e2), "(2e-6) dx", np.abs(s.coords - s2.coords).max(), "(5e-5) dv", np.abs(s.velocities - s2.velocities).max(), "(8e-3)")
for kind in ("lj_fp32_packed", "pme_fp32"):
    case, dtype = make(kind)
    s = case.system(pkg, dtype)
    everything(pkg, s)
    pkg.scale_coords(s, np.diag([1.02, 0.97, 1.015]))
    f1, e1, w1, nl1 = everything(pkg, s)
    s2 = on_box(case, dtype, s.coords, s.boundary.side_lengths).system(pkg, dtype)
    f2, e2, w2, nl2 = everything(pkg, s2)
    print(kind, "df/fmax", np.abs(f1 - f2).max() / np.abs(f2).max(), "de/e", abs(e1 - e2) / abs(e2), "dw/w", np.abs(w1 - w2).max() / np.abs(w2).max(), "(bar 2e-5 each)")

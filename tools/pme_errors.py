"""Print the achieved PME parity numbers on the GPU (diagnostic)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader; m = molly_loader.load()
from tests import golden6mrr as G
d = G.data()
case = G.case("ewald", np.float64, bonded=True, approx_erfc=False, pme=True)
s = case.system(m, np.float64)
f = m.forces(s)
print("fp64 all_pme_exact max|df| vs OpenMM", np.linalg.norm(f - d["openmm_forces_all_pme_exact"], axis=1).max())
e = m.potential_energy(s) + G.lj_dispersion_correction(d)
print("fp64 energy diff", e - float(d["openmm_energy_all_pme_exact"]))
m.simulate(s, m.VelocityVerlet(dt=0.0005), 100)
xo = d["openmm_coordinates_100steps"]; box = case.box
dx = s.coords - (xo - np.floor(xo / box) * box); dx -= np.round(dx / box) * box
print("100 steps max|dx|", np.linalg.norm(dx, axis=1).max(), "max|dv|", np.linalg.norm(s.velocities - d["openmm_velocities_100steps"], axis=1).max())
case32 = G.case("ewald", np.float32, bonded=True, approx_erfc=True, pme=True)
s32 = case32.system(m, np.float32)
f32 = m.forces(s32).astype(np.float64)
err = np.linalg.norm(f32 - d["openmm_forces_all_pme"], axis=1)
print("fp32 all_pme max|df| vs OpenMM", err.max(), "rms", np.sqrt((err**2).mean()), "force rms", np.sqrt((d["openmm_forces_all_pme"]**2).sum(1).mean()))

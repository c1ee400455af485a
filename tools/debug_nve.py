import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader; m = molly_loader.load()
from tests import golden6mrr as G
case = G.case("rf", np.float64, bonded=True)
s = case.system(m, np.float64)
for k in range(11):
    ke = m.kinetic_energy(s); pe = m.potential_energy(s)
    print(k * 20, "KE %.4f PE %.4f E %.4f" % (ke, pe, ke + pe))
    m.simulate(s, m.VelocityVerlet(dt=0.0005, remove_CM_motion=0), 20, init_step=20 * k)

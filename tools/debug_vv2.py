import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader; m = molly_loader.load()
from tests import systems as S
case = S.charged_fluid(10, dict(kind="rf", rc=1.0), dtype=np.float64, with_exceptions=True)
s = case.system(m, np.float64); m.simulate(s, m.VelocityVerlet(dt=0.0005), 40)

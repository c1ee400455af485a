"""debug helper: the host-driven set_state → forces(step_n) loop of tests/test_gpu_cadence.py with the engine's list decisions printed"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MOLLYHIP_DEBUG", "1")
import molly_loader
from tests import systems as S
pkg = molly_loader.load()
dtype = np.float32
case = S.charged_fluid(14, dict(kind="rf", rc=1.0), dtype=dtype, stable=True)
s = case.system(pkg, dtype)
o = case.oracle(np.float64)
for step in range(0, 31):
    if step:
        o.vv_run(1, 0.0005, first_step=step - 1, remove_cm_every=1)
    s.coords[:] = o.coords.astype(dtype)
    pkg.forces(s, step_n=step)
    st = s.stats()
    print("step", step, "outer", st["n_outer_builds"], "prunes", st["n_filter_passes"], "rebuild events", st["n_rebuilds"], "minimg", st["minimg_mode"], flush=True)

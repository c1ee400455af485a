import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import molly_loader
from tests import systems as S
m = molly_loader.load()
for n_side in (16, 40):
    case = S.lj_fluid(n_side, dtype=np.float32)
    o = case.oracle(np.float64); f_ref = o.forces(o.neighbors("cell", nthreads=8), nthreads=8)
    for lean in ("0", "1"):
        os.environ["MOLLYHIP_PRUNE_LEAN"] = lean
        s = case.system(m, np.float32)
        f = m.forces(s).astype(np.float64)
        st = s.stats()
        err = np.linalg.norm(f - f_ref, axis=1)
        print(n_side, "lean", lean, "max err", err.max(), "rel", err.max() / np.linalg.norm(f_ref, axis=1).max(), "slots", st["n_list_slots"], "tile", st["max_tile_atoms"], "BI", st["block_atoms"], st["j_split"], "bad atoms", int((err > 1e-2).sum()))
        s.close()

import sys, os, ctypes as C, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader; m = molly_loader.load()
from tests import systems as S
arg = sys.argv[1] if len(sys.argv) > 1 else "64"
if arg == "6mrr":
    from tests import golden6mrr
    case = golden6mrr.case("ewald", np.float32, bonded=False)
else:
    case = S.lj_fluid(int(arg), dtype=np.float32)
s = case.system(m, np.float32)
s.push_state()
L = m.lib(); ctx = s.engine()
s._check(L.mhip_rebuild(ctx, 0))
s._check(L.mhip_set_profiling(ctx, 1))
for k in range(10):
    s.push_state()                       # marks the list stale: every call is a full (outer) search
    s._check(L.mhip_rebuild(ctx, k + 1))
st = s.stats()
print("debug", os.environ.get("MOLLYHIP_BUILD_DEBUG", "0"), "build kernel ms", st["prof_ms"][1] / max(st["prof_calls"][1], 1), "sort ms", st["prof_ms"][3] / max(st["prof_calls"][3], 1),
      "max_tile", st["max_tile_atoms"], "avg tile", st["tile_atoms_total"] / st["n_blocks"], "lds", st["lds_bytes"])

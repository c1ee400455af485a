import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader; m = molly_loader.load()
from tests import systems as S
case = S.lj_fluid(64, dtype=np.float32)
o = case.oracle(np.float32)
oi, oj, osp = o.neighbors("cell", nthreads=16)
s = case.system(m, np.float32)
nl = m.find_neighbors(s)
a = set(zip(np.minimum(oi, oj).tolist(), np.maximum(oi, oj).tolist()))
b = set(zip(nl.i.tolist(), nl.j.tolist()))
print("oracle", len(a), "gpu", len(b), "gpu list entries", nl.n)
miss = sorted(a - b)[:10]; extra = sorted(b - a)[:10]
print("missing", miss, "extra", extra)
x = case.coords.astype(np.float32)
box = np.float32(case.box[0])
for (i, j) in miss + extra:
    d = x[j] - x[i]
    d64 = d.astype(np.float64); d64 -= np.round(d64 / case.box) * case.box
    print(i, j, x[i], x[j], "r2(64)=%.9f r_list2=%.9f" % ((d64**2).sum(), 1.2**2), "r2(f32 of 1.2^2)=%.9f" % (np.float32(1.2)*np.float32(1.2)))
st = s.stats(); print(st)

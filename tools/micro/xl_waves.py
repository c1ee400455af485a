"""Which i-waves of a block lose their neighbours in the single-list search with exception lists (diagnostic):  python tools/micro/xl_waves.py N_SIDE [ENV=V ...]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
for kv in sys.argv[2:]:
    k, v = kv.split("="); os.environ[k] = v
os.environ.setdefault("MOLLYHIP_OUTER_MARGIN_PM", "0")
import molly_loader
pkg = molly_loader.load()
from tests import systems as S
n_side = int(sys.argv[1])
DT = np.float64 if os.environ.get("XLW_DTYPE") == "f64" else np.float32
case = S.charged_fluid(n_side, dict(kind="rf", rc=1.0, weight_special=0.8333333333333334), dtype=DT, stable=True)
s = case.system(pkg, DT)
nl = pkg.find_neighbors(s)
st = s.stats()
L = pkg.lib()
perm = np.empty(case.n, np.int32)
s._check(L.mhip_export_order(s.engine(), s._ptr(perm), case.n))       # perm[sorted slot] = caller index
cnt = np.bincount(nl.i, minlength=case.n) + np.bincount(nl.j, minlength=case.n)
per_slot = cnt[perm]
bi = st["block_atoms"]
oi, oj, _ = case.oracle(DT).neighbors("cell", nthreads=16)
ref = (np.bincount(oi, minlength=case.n) + np.bincount(oj, minlength=case.n))[perm]
print(f"shape {bi}x{st['j_split']}, pairs {nl.n} of {len(oi)}")
for b in (0, 1, 7, 100):
    sl = slice(b * bi, (b + 1) * bi)
    print(f"block {b}: neighbours per atom by 64-lane wave, found / expected:", [(int(per_slot[sl][w * 64:(w + 1) * 64].sum()), int(ref[sl][w * 64:(w + 1) * 64].sum())) for w in range(bi // 64)])
w_all = per_slot.reshape(-1)[: (case.n // bi) * bi].reshape(-1, bi // 64, 64).sum(axis=(0, 2))
r_all = ref.reshape(-1)[: (case.n // bi) * bi].reshape(-1, bi // 64, 64).sum(axis=(0, 2))
print("all blocks, by wave:", list(zip(w_all.tolist(), r_all.tolist())))

"""debug: the triclinic single list with exceptions beyond 64-atom blocks (tests/test_gpu_triclinic.py) — where do the forces differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MOLLYHIP_OUTER_MARGIN_PM"] = os.environ.get("MOLLYHIP_OUTER_MARGIN_PM", "0")
from tests import systems as S
from tests.test_gpu_triclinic import sheared_fluid
import molly_loader
pkg = molly_loader.load()
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 35
with_xl = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
basis, case = sheared_fluid(np.float32, n_side=n_side, jitter=float(sys.argv[3]) if len(sys.argv) > 3 else 0.008)
if with_xl:
    idx = np.arange(case.n - 2)
    case.excluded = np.stack([idx[idx % 3 == 0], idx[idx % 3 == 0] + 1], 1)
    case.special = np.stack([idx[idx % 3 == 1], idx[idx % 3 == 1] + 2], 1)
    case.lj = dict(cutoff=("distance", 1.0), weight_special=0.5)
s = case.system(pkg, np.float32)
got = pkg.find_neighbors(s)
ref = case.oracle(np.float32).neighbors("brute", nthreads=16)
same = got.n == len(ref[0]) and all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(got.i, got.j, got.special), S.sorted_pairs(*ref)))
print("pair set + special flags equal to the fp32 brute-force oracle:", same, got.n, len(ref[0]))
o = case.oracle(np.float64)
nl = o.neighbors("brute", nthreads=16)
f_ref = o.forces(nl, nthreads=8)
scale, jump = o.force_scale(nl)
f = pkg.forces(s).astype(np.float64)
err = np.linalg.norm(f - f_ref, axis=1); tol = 6e-5 * scale + 1.01 * jump + 1e-4
bad = np.nonzero(err > tol)[0]
st = s.stats()
print("n", case.n, "xl", with_xl, "block", st["block_atoms"], st["j_split"], "minimg", st["minimg_mode"], "bad atoms", len(bad), "worst ratio", (err / tol).max(), "rel rms", S.rel_rms(err, f_ref))
o32 = case.oracle(np.float32)
f32 = o32.forces(o32.neighbors("brute", nthreads=16), nthreads=1).astype(np.float64)
e32 = np.linalg.norm(f32 - f_ref, axis=1)
print("the reference's arithmetic in fp32: worst ratio", (e32 / tol).max(), "bad", int((e32 > tol).sum()), "rel rms", S.rel_rms(e32, f_ref))
for a in bad[:10]:
    print(a, "err", err[a], "tol", tol[a], "scale", scale[a], "|f|", np.linalg.norm(f_ref[a]), "fp32-oracle err", e32[a])

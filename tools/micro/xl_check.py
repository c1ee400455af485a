"""A charged fluid WITH exception lists at the sizes where the blocks are 128 or 256 atoms (diagnostic): pair count and fp32 forces against the oracle,
by search variant.   python tools/micro/xl_check.py N_SIDE"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    n_side = int(sys.argv[2])
    import molly_loader
    pkg = molly_loader.load()
    from tests import systems as S
    case = S.charged_fluid(n_side, dict(kind="rf", rc=1.0, weight_special=0.8333333333333334), dtype=np.float32, stable=True)
    s = case.system(pkg, np.float32)
    if os.environ.get("TOOL_SHAPE"):      # (the launch shape through the API: mhip_set_launch_config)
        bi, js = map(int, os.environ["TOOL_SHAPE"].split("x")); pkg.set_launch_config(s, bi, js)
    f = pkg.forces(s).astype(np.float64)
    nl = pkg.find_neighbors(s)
    st = s.stats()
    np.save(sys.argv[3], f)
    print(json.dumps({"block": st["block_atoms"], "js": st["j_split"], "tile": st["max_tile_atoms"], "pairs": int(nl.n), "special": int(np.asarray(nl.special).sum())}))
    sys.exit(0)

n_side = int(sys.argv[1])
from tests import systems as S
case = S.charged_fluid(n_side, dict(kind="rf", rc=1.0, weight_special=0.8333333333333334), dtype=np.float32, stable=True)
o32 = case.oracle(np.float32)
oi, oj, osp = o32.neighbors("cell", nthreads=16)
o = case.oracle(np.float64)
nl = o.neighbors("cell", nthreads=16)
f_ref = o.forces(nl, nthreads=16)
scale = np.linalg.norm(f_ref, axis=1).mean()
print(f"n_side {n_side}: {case.n} atoms, oracle {len(oi)} pairs, {int(np.asarray(osp).sum())} special")
for name, env in [("default", {}), ("64 x 16", {"TOOL_SHAPE": "64x16"}), ("single list", {"MOLLYHIP_OUTER_MARGIN_PM": "0"})]:
    e = dict(os.environ); e.update(env)
    out = f"/tmp/xl_{os.getpid()}.npy"
    r = subprocess.run([sys.executable, __file__, "--child", str(n_side), out], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"{name}: FAILED {r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else r.returncode}"); continue
    err = np.linalg.norm(np.load(out) - f_ref, axis=1)
    print(f"   {name}: {line[0]} | max |df| {err.max():.3e} (mean |f| {scale:.3e}), {int((err > 1e-3 * scale).sum())} atoms off")

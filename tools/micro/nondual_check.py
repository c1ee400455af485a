"""Single-domain forces with the single (non-dual) pair list against the default dual list, by block shape (diagnostic):
python tools/micro/nondual_check.py N_SIDE [ENV=VALUE ...]"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    n_side = int(sys.argv[2])
    import molly_loader
    pkg = molly_loader.load()
    from tests import systems as S
    case = S.lj_fluid(n_side, dtype=np.float32)
    s = case.system(pkg, np.float32)
    if os.environ.get("TOOL_SHAPE"):      # (the launch shape through the API: mhip_set_launch_config)
        bi, js = map(int, os.environ["TOOL_SHAPE"].split("x")); pkg.set_launch_config(s, bi, js)
    f = pkg.forces(s).astype(np.float64)
    st = s.stats()
    np.save(sys.argv[3], f)
    print(json.dumps({"block": st["block_atoms"], "js": st["j_split"], "tile": st["max_tile_atoms"], "pairs": st["n_pairs_full"]}))
    sys.exit(0)

n_side = int(sys.argv[1])
variants = [("dual (default)", {}), ("single list, walk", {"MOLLYHIP_OUTER_MARGIN_PM": "0"}), ("single list, walk, 64x16", {"MOLLYHIP_OUTER_MARGIN_PM": "0", "TOOL_SHAPE": "64x16"})]
ref = None
for name, env in variants:
    e = dict(os.environ); e.update(env)
    for kv in sys.argv[2:]:
        k, v = kv.split("="); e[k] = v
    out = f"/tmp/nd_{os.getpid()}.npy"
    r = subprocess.run([sys.executable, __file__, "--child", str(n_side), out], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"{name}: FAILED {r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else r.returncode}"); continue
    f = np.load(out)
    if ref is None:
        ref = f
    err = np.linalg.norm(f - ref, axis=1)
    print(f"n_side {n_side} {name}: {line[0]} | against the dual list: max |df| {err.max():.3e} (mean |f| {np.linalg.norm(ref, axis=1).mean():.3e}), {int((err > 1e-3 * np.linalg.norm(ref, axis=1).mean()).sum())} atoms off")

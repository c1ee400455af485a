"""Timing experiment: where the outer neighbour search (k_build) spends its time at 1M atoms.  MOLLYHIP_BUILD_DEBUG=n makes the
kernel return after stage n (1 boxes, 2 cell pruning + scan, 3 tile staging, 8 = all but the i-loop of the search, 7 = search without
the per-lane unpacking / emission, 4 = everything but the final padding); the lists are unusable then, only the time is read."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, ctypes as C
    import molly_loader
    from tests import systems as S
    m = molly_loader.load()
    if sys.argv[1] == "6mrr":
        from tests import golden6mrr
        case = golden6mrr.case("ewald", dtype=np.float32, bonded=False, pme=False)
    else:
        case = S.lj_fluid(int(sys.argv[1]), seed=4, dtype=np.float32)
    s = case.system(m, np.float32)
    s.push_state(velocities=True)
    L = m.lib(); ctx = s.engine()
    s._check(L.mhip_set_profiling(ctx, 1))
    for k in range(3):
        s._check(L.mhip_set_state(ctx, s._ptr(s.coords), None, 0))
        os.environ["MOLLYHIP_SET_STATE_REBUILDS"] = "1"
        try:
            s._check(L.mhip_rebuild(ctx, 10 * k))
        except Exception as e:
            print("rebuild error", e)
    st = s.stats() if os.environ.get("MOLLYHIP_BUILD_DEBUG", "0") == "0" else None
    import molly_jl_amd._lib as _lib
    stt = _lib.Stats(); L.mhip_get_stats(ctx, C.byref(stt))
    print(json.dumps({"debug": os.environ.get("MOLLYHIP_BUILD_DEBUG", "0"), "build_ms": stt.prof_ms[1] / max(stt.prof_calls[1], 1), "calls": stt.prof_calls[1],
                      "sort_ms": stt.prof_ms[3] / max(stt.prof_calls[3], 1), "max_tile": stt.max_tile_atoms, "BI": stt.block_atoms, "JS": stt.j_split}))
else:
    for dbg in ("0", "1", "2", "3", "8", "7", "4"):
        env = dict(os.environ, MOLLYHIP_BUILD_DEBUG=dbg, MOLLYHIP_SET_STATE_REBUILDS="1")
        r = subprocess.run([sys.executable, __file__, os.environ.get("BUILD_BREAKDOWN_CASE", "100")], env=env, capture_output=True, text=True)
        print(r.stdout.strip().split("\n")[-1] if r.stdout.strip() else ("ERR " + r.stderr[-300:]))

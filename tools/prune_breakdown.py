"""Timing experiment: cost structure of the list filter (k_filter: outer list → list of radius r_list with a compacted tile) at 1M atoms.
MOLLYHIP_FILTER_DEBUG: 0 complete, 1 no row stores, 2 + no LDS marks, 3 + no tile compaction / renumbering.  Only times are read."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, ctypes as C
    import molly_loader
    from tests import systems as S
    m = molly_loader.load()
    case = S.lj_fluid(int(sys.argv[1]), seed=4, dtype=np.float32)
    s = case.system(m, np.float32)
    s.push_state(velocities=True)
    L = m.lib(); ctx = s.engine()
    s._check(L.mhip_vv_run(ctx, 0, 25, 0.002, 1))
    s._check(L.mhip_set_profiling(ctx, 1))
    n = C.c_int64(0)
    for k in range(4):
        s._check(L.mhip_export_neighbors(ctx, None, None, None, 0, C.byref(n)))
    import molly_jl_amd._lib as _lib
    stt = _lib.Stats(); L.mhip_get_stats(ctx, C.byref(stt))
    print(json.dumps({"debug": os.environ.get("MOLLYHIP_FILTER_DEBUG", "0"), "filter_ms": stt.prof_ms[4] / max(stt.prof_calls[4], 1), "calls": stt.prof_calls[4], "pairs": n.value}))
else:
    for dbg in ("0", "1", "2", "3"):
        env = dict(os.environ, MOLLYHIP_FILTER_DEBUG=dbg)
        r = subprocess.run([sys.executable, __file__, "100"], env=env, capture_output=True, text=True)
        print(r.stdout.strip().split("\n")[-1] if r.stdout.strip() else ("ERR " + r.stderr[-300:]))

#!/usr/bin/env python3
"""A/B timing of builds of libmollyhip.so on one GPU box: per-stage hipEvent times of the 1M-atom (or another) workload.

    python tools/force_ab.py [--workload lj1m] [--steps 600] lib_a.so lib_b.so ...     (a path of "-" = the in-tree build)

Each library runs in its own process (MOLLYHIP_LIB_AB), equilibrates, then times `--steps` steps with the stage timers on.
Extra environment for a run: NAME=VALUE pairs joined by commas after the path, e.g.  lib.so:MOLLYHIP_INNER_SKIN_PM=120,MOLLYHIP_X=1
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(workload, steps, equil, static=False):
    sys.path.insert(0, ROOT)
    import time
    import bench
    import molly_loader
    m = molly_loader.load()
    case, dtype, dt = bench.make_case(workload)
    L = m.lib()
    s = case.system(m, dtype)
    s.push_state(velocities=True)
    ctx = s.engine()
    if static:   # frozen coordinates: only force passes (timing a force pass on its own)
        import numpy as np
        import torch
        f = torch.empty((case.n, 3), dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
        call = lambda: s._check(L.mhip_forces(ctx, 1, 0, f.data_ptr(), None, 1))
        for _ in range(5):
            call()
        s._check(L.mhip_set_profiling(ctx, 1))
        for _ in range(steps):
            call()
        st = s.stats()
        print("AB_RESULT " + json.dumps({"ms_per_step": 0.0, "per_call_us": {"forces": 1e3 * st["prof_ms"][0] / max(st["prof_calls"][0], 1)}, "per_step_us": {}, "calls": {},
                                         "n_list_slots": st["n_list_slots"], "max_tile": st["max_tile_atoms"], "lds": st["lds_bytes"], "n_outer": st["n_outer_builds"], "n_prunes": st["n_filter_passes"]}))
        return
    run = lambda first, n: s._check(L.mhip_vv_run(ctx, first, n, dt, 1))
    if os.environ.get("AB_STATS_FIRST"):
        s._check(L.mhip_rebuild(ctx, 0)); s.stats()
    if os.environ.get("AB_ALLOC_GB"):
        import torch
        keep = torch.empty(int(float(os.environ["AB_ALLOC_GB"]) * 2**30), dtype=torch.uint8, device="cuda"); keep.zero_()
    if equil:
        run(0, equil)
    run(equil, 200)
    first = equil + 200
    s._check(L.mhip_synchronize(ctx))
    t0 = time.perf_counter()
    run(first, steps)
    s._check(L.mhip_synchronize(ctx))
    ms = (time.perf_counter() - t0) * 1e3 / steps
    if os.environ.get("AB_STATS_BEFORE"):
        s.stats()
    s._check(L.mhip_set_profiling(ctx, 1))
    run(first + steps, steps)
    st = s.stats()
    s._check(L.mhip_set_profiling(ctx, 0))
    s._check(L.mhip_check_finite(ctx))
    names = ["forces", "build", "integrator", "sort", "prune", "bonded", "pme", "x"]
    out = {"ms_per_step": ms, "per_call_us": {n: 1e3 * st["prof_ms"][k] / max(st["prof_calls"][k], 1) for k, n in enumerate(names) if st["prof_calls"][k]},
           "per_step_us": {n: 1e3 * st["prof_ms"][k] / steps for k, n in enumerate(names) if st["prof_calls"][k]},
           "calls": {n: st["prof_calls"][k] for k, n in enumerate(names) if st["prof_calls"][k]},
           "n_list_slots": st["n_list_slots"], "max_tile": st["max_tile_atoms"], "lds": st["lds_bytes"], "n_outer": st["n_outer_builds"], "n_prunes": st["n_filter_passes"]}
    print("AB_RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="lj1m")
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--equil", type=int, default=None)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--static", action="store_true", help="frozen start coordinates, force passes only")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    equil = a.equil if a.equil is not None else (2000 if a.workload.startswith("lj") else 0)
    if a.child:
        return child(a.workload, a.steps, equil, a.static)
    for spec in a.libs or ["tree"]:
        path, _, envs = spec.partition(":")
        env = dict(os.environ)
        if path not in ("-", "tree"):
            env["MOLLYHIP_LIB_AB"] = os.path.abspath(path)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--workload", a.workload, "--steps", str(a.steps), "--equil", str(equil)] + (["--static"] if a.static else []),
                           env=env, capture_output=True, text=True, timeout=1200)
        res = [l for l in r.stdout.splitlines() if l.startswith("AB_RESULT ")]
        if not res:
            print(f"{spec}: FAILED rc {r.returncode}\n{r.stdout[-800:]}\n{r.stderr[-1500:]}")
            continue
        d = json.loads(res[0][10:])
        print(f"{spec}: {d['ms_per_step']:.4f} ms/step | per call us: " + " ".join(f"{k} {v:.1f}" for k, v in d["per_call_us"].items()) +
              " | per step us: " + " ".join(f"{k} {v:.1f}" for k, v in d["per_step_us"].items()) +
              f" | outer {d['n_outer']} prunes {d['n_prunes']} slots {d['n_list_slots']} max_tile {d['max_tile']} lds {d['lds']}", flush=True)


if __name__ == "__main__":
    main()

"""The reference README's GPU example (README.md:100-137), call by call, through libmollyhip.so on one MI355X:

    sys = System("6mrr_equil.pdb", ff; nonbonded_method=:pme, array_type=CuArray)       → the 6mrr inputs of molly.jl_amd/data (fixture builder: tools/param_6mrr.py)
    simulate!(sys, SteepestDescentMinimizer())                                          → simulate(sys, SteepestDescentMinimizer())
    random_velocities!(sys, 298 K)                                                      → random_velocities(sys, 298.0)
    simulate!(sys, Langevin(dt=0.001 ps, T, friction=1/ps, coupling=MonteCarloBarostat(1 bar, T, sys.boundary)), 5_000)

What differs from the README: no loggers / trajectory file (IO is outside the engine's scope), and the barostat scales every atom's coordinates (the branch of
scale_coords! for systems without a topology, spatial.jl:1198-1209) — the rigid-molecule branch is host code that stays in Julia — so the flexible bonds take part in
each trial and the volume moves are small.  Prints one JSON line.      python examples/readme_6mrr_npt.py [--steps 5000] [--minimize-steps 100]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--minimize-steps", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    m = molly_loader.load()
    import importlib
    W = importlib.import_module("molly_jl_amd.workloads")
    T = np.float32
    case = W.protein_6mrr("ewald", T, pme=True)
    s = case.system(m, T)
    rng = np.random.default_rng(args.seed)
    e0 = m.potential_energy(s)
    t0 = time.perf_counter()
    m.simulate(s, m.SteepestDescentMinimizer(max_steps=args.minimize_steps))
    t_min = time.perf_counter() - t0
    e1 = m.potential_energy(s)
    temp = 298.0
    m.random_velocities(s, temp, rng=rng)
    baro = m.MonteCarloBarostat(1.0, temp, s.boundary)
    sim = m.Langevin(dt=0.001, temperature=temp, friction=1.0, coupling=baro)
    v0 = m.volume(s.boundary)
    n_box0 = s.stats()["n_box_changes"]
    t0 = time.perf_counter()
    m.simulate(s, sim, args.steps, rng=rng)
    t_run = time.perf_counter() - t0
    st = s.stats()
    out = {"example": "README.md:100-137 (6mrr, PME, SteepestDescentMinimizer, Langevin 1 fs + MonteCarloBarostat 1 bar)", "n_atoms": len(s), "dtype": "f32",
           "minimize_steps": args.minimize_steps, "minimize_s": round(t_min, 3), "pe_before_kj_mol": round(e0, 1), "pe_minimized_kj_mol": round(e1, 1),
           "md_steps": args.steps, "md_s": round(t_run, 3), "ms_per_step_incl_barostat": round(1e3 * t_run / max(args.steps, 1), 4),
           "ns_per_day": round(86400.0 * 1e-6 * args.steps / t_run, 1) if args.steps else None,
           "barostat_trials": args.steps // baro.n_steps, "box_changes": st["n_box_changes"] - n_box0, "volume_nm3": [round(v0, 4), round(m.volume(s.boundary), 4)],
           "temperature_K": round(m.temperature(s), 1), "pe_final_kj_mol": round(m.potential_energy(s), 1), "finite": bool(np.isfinite(s.coords).all())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

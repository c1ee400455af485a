"""SURVEY §8(d) cfg 5: 6mrr reaction-field fp64 NVE, remove_CM_motion = 0, dt = 0.5 fs, 20 000 steps, KE + PE every 100 steps.
Reports max |E − E0| and the least-squares linear drift per ns per atom (report, not a gate).  Also the LJ fluid of
test/energy_conservation.jl's kind (2000+ atoms, fp64).  Needs an MI355X:   python tools/nve_drift.py > gpurun_out/nve_drift.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molly_loader  # noqa: E402

m = molly_loader.load()
UPKEEP = []


def run(case, dtype, dt, n_steps, every, label):
    s = case.system(m, dtype)
    sim = m.VelocityVerlet(dt=dt, remove_CM_motion=0)
    es = [m.total_energy(s)]
    for k in range(n_steps // every):
        m.simulate(s, sim, every, init_step=k * every)
        es.append(m.total_energy(s))
    st = s.stats()
    UPKEEP.append({"outer_searches": st["n_outer_builds"], "prunes": st["n_filter_passes"], "steps": n_steps})
    es = np.array(es)
    t_ns = np.arange(len(es)) * every * dt * 1e-3
    slope = np.polyfit(t_ns, es, 1)[0]
    return {"system": label, "n_atoms": case.n, "dtype": np.dtype(dtype).name, "dt_fs": dt * 1e3, "steps": n_steps, "log_every": every,
            "E0_kJ_mol": float(es[0]), "max_abs_dE_kJ_mol": float(np.abs(es - es[0]).max()), "rms_dE_kJ_mol": float(np.std(es)),
            "max_abs_dE_second_half_vs_mean": float(np.abs(es[len(es) // 2:] - es[len(es) // 2:].mean()).max()),
            "linear_drift_kJ_mol_per_ns_per_atom": float(slope / case.n),
            "kT_300K_kJ_mol": 2.494, "temperature_end_K": float(m.temperature(s))}


if __name__ == "__main__":
    from tests import golden6mrr as G
    from tests import systems as S
    out = [run(G.case("rf", np.float64, bonded=True), np.float64, 0.0005, 20000, 100, "6mrr reaction field + LJ + bonded (BASELINE configs[4])"),
           run(S.lj_fluid(14, dtype=np.float64), np.float64, 0.002, 20000, 100, "LJ fluid 2744 atoms (test/energy_conservation.jl kind)")]
    if "--lj1m" in sys.argv:   # the benchmark system itself, fp32, after equilibration: what the list upkeep (inner skin, extra checks) does to the energy
        case = S.lj_fluid(100, dtype=np.float32)
        sy = case.system(m, np.float32)
        m.simulate(sy, m.VelocityVerlet(dt=0.002, remove_CM_motion=1), 2000)
        case.coords, case.velocities = np.array(sy.coords, dtype=np.float64), np.array(sy.velocities, dtype=np.float64)
        out.append(run(case, np.float32, 0.002, 10000, 500, "1M-atom LJ fluid, fp32, equilibrated 2000 steps first (bench.py's lj1m)"))
        # the list upkeep the energy figure was obtained with — and a guard: a list that silently over-searches is a performance
        # regression (one outer search per ≈ 100 steps, one prune per ≈ 25 at 85 K, dt 2 fs)
        out[-1]["list_upkeep"] = UPKEEP[-1]
        assert UPKEEP[-1]["outer_searches"] <= 10000 // 60 and UPKEEP[-1]["prunes"] <= 10000 // 15, UPKEEP[-1]
    print(json.dumps(out, indent=1))

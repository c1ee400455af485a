"""SURVEY §8(d) cfg 5: 6mrr reaction-field fp64 NVE, remove_CM_motion = 0, dt = 0.5 fs, 20 000 steps, KE + PE every 100 steps.
Reports max |E − E0| and the least-squares linear drift per ns per atom (report, not a gate).  Also the LJ fluid of
test/energy_conservation.jl's kind (2000+ atoms, fp64).  Needs an MI355X:   python tools/nve_drift.py > gpurun_out/nve_drift.json
`--oracle [N]`: instead, N (2000) steps of the 6mrr system on the engine AND on the CPU oracle, both energy traces side by side."""
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


def oracle_trace(case, dt, n_steps, every, nthreads):
    """The ORACLE's energy trace (CPU restatement of the reference, fp64) over the same steps of the same system, next to the engine's: if
    both show the same excursions and the same drift, the drift is the physics of the system and the time step (flexible TIP3P from a
    constrained-equilibrated start, velocity Verlet at ω·dt ≈ 0.35), not a defect of the engine."""
    o = case.oracle(np.float64)
    o.native = True
    specific = case.bonds is not None
    def energy():
        nl = o.neighbors("cell", nthreads=nthreads)
        return o.kinetic_energy() + o.potential_energy(nl, specific=specific)
    es = [energy()]
    for k in range(n_steps // every):
        o.vv_run(every, dt, first_step=k * every, remove_cm_every=0, nthreads=nthreads, specific=specific)
        es.append(energy())
    return np.array(es)


def engine_trace(case, dtype, dt, n_steps, every):
    s = case.system(m, dtype)
    sim = m.VelocityVerlet(dt=dt, remove_CM_motion=0)
    es = [m.total_energy(s)]
    for k in range(n_steps // every):
        m.simulate(s, sim, every, init_step=k * every)
        es.append(m.total_energy(s))
    return np.array(es)


def compare_with_oracle(n_steps=2000, every=100):
    from tests import golden6mrr as G
    case = G.case("rf", np.float64, bonded=True)
    nthreads = min(os.cpu_count() or 1, 64)
    e_orc = oracle_trace(case, 0.0005, n_steps, every, nthreads)
    e_eng = engine_trace(case, np.float64, 0.0005, n_steps, every)
    t_ns = np.arange(len(e_orc)) * every * 0.0005 * 1e-3
    return {"system": "6mrr reaction field + LJ + bonded (BASELINE configs[4]): engine against the oracle, same start, same steps",
            "n_atoms": case.n, "steps": n_steps, "log_every": every, "dt_fs": 0.5, "oracle_threads": nthreads,
            "E_engine_kJ_mol": [float(v) for v in e_eng], "E_oracle_kJ_mol": [float(v) for v in e_orc],
            "max_abs_E_engine_minus_E_oracle_kJ_mol": float(np.abs(e_eng - e_orc).max()),
            "max_abs_dE_engine_kJ_mol": float(np.abs(e_eng - e_eng[0]).max()), "max_abs_dE_oracle_kJ_mol": float(np.abs(e_orc - e_orc[0]).max()),
            "linear_drift_engine_kJ_mol_per_ns_per_atom": float(np.polyfit(t_ns, e_eng, 1)[0] / case.n),
            "linear_drift_oracle_kJ_mol_per_ns_per_atom": float(np.polyfit(t_ns, e_orc, 1)[0] / case.n)}


if __name__ == "__main__":
    if "--oracle" in sys.argv:      # only the engine-against-oracle comparison (≈ 1 min of CPU at 64 threads)
        k = sys.argv.index("--oracle")
        n = int(sys.argv[k + 1]) if len(sys.argv) > k + 1 and sys.argv[k + 1].isdigit() else 2000
        print(json.dumps([compare_with_oracle(n)], indent=1))
        sys.exit(0)
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

#!/usr/bin/env python3
"""bench.py — ns/day (and Matom-steps/s) of the MI355X nonbonded engine on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W [--workload lj1m|lj256k|6mrr_pme|6mrr_rf64]

One "step" = one velocity-Verlet MD step (forces + integration + amortised neighbour rebuild) of the whole
system, state resident in HBM.  N = 1: the whole box on one GPU.  N > 1 (launched by torch.distributed.run,
one rank per GPU): the same box cut into spatial bricks with RCCL ghost-coordinate exchange — total work is
fixed, so scaling is "strong".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def make_case(workload):
    from tests import systems as S
    if workload == "lj1m":
        return S.lj_fluid(100, seed=4, dtype=np.float32), np.float32, 0.002
    if workload == "lj256k":
        return S.lj_fluid(64, seed=2, dtype=np.float32), np.float32, 0.002
    if workload in ("6mrr_pme", "6mrr_direct", "6mrr_rf64"):
        from tests import golden6mrr
        dtype = np.float64 if workload == "6mrr_rf64" else np.float32
        return golden6mrr.case("rf" if workload == "6mrr_rf64" else "ewald", dtype=dtype, bonded=True, pme=(workload == "6mrr_pme")), dtype, 0.0005
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(case, dtype, dt, budget_s=20.0):
    """The Molly-algorithm CPU restatement (oracle, kind "port") timed on this host's cores on a bounded
    sample of the same workload: same atoms, same parameters, a handful of steps."""
    from oracle import pyoracle as orc
    orc.build(native=True)
    cores = os.cpu_count() or 1
    nthreads = min(cores, 64)
    o = case.oracle(dtype)
    o.native = True
    specific = case.bonds is not None
    general = case.pme is not None
    t0 = time.perf_counter()
    o.vv_run(1, dt, nthreads=nthreads, specific=specific, general=general)   # one step incl. the initial neighbour build + force pass
    t1 = time.perf_counter() - t0
    # steady-state sample: as many steps as fit the budget, at least one rebuild interval if affordable
    n = int(max(2, min(20, budget_s / max(t1 / 2.0, 1e-3))))
    t0 = time.perf_counter()
    o.vv_run(n, dt, first_step=1, nthreads=nthreads, specific=specific, general=general)
    t = time.perf_counter() - t0
    steps_s = n / t
    # the same algorithm on ONE core (the reference's 1-thread pair loop, src/force.jl:828-884): a couple of steps that avoid the
    # rebuild cadence (the list of the run above is rebuilt at the first step of a run, so that cost is reported on its own)
    n1 = int(max(1, min(4, (budget_s / 2.0) / max(t / n * nthreads * 0.6, 1e-3))))
    t0 = time.perf_counter()
    o.vv_run(n1, dt, first_step=n + 1, nthreads=1, specific=specific, general=general)
    t1c = time.perf_counter() - t0
    steps_s1 = n1 / t1c
    return {"value": steps_s * dt * 1e3 * 86400 * 1e-6, "unit": "ns/day", "cores": nthreads, "kind": "port",
            "matom_steps_per_s": steps_s * case.n / 1e6,
            "sample": f"{n} velocity-Verlet steps of the full {case.n}-atom system (threaded pair loop of src/force.jl:886-969 + "
                      f"cell-list rebuild every {case.rebuild_every} steps" + (", PME reciprocal space on ONE thread (the restatement's mesh code is serial)" if general else "") + f"), {nthreads} threads, -O3 -march=native",
            "one_core": {"value": steps_s1 * dt * 1e3 * 86400 * 1e-6, "unit": "ns/day", "cores": 1, "matom_steps_per_s": steps_s1 * case.n / 1e6,
                         "sample": f"{n1} step(s) incl. the neighbour search at the start of the run (1-thread pair loop of src/force.jl:828-884), same system"}}


def load_traffic(workload):
    """HBM bytes per force-kernel launch from the rocprofv3 PMC passes committed under profiles/ (collected by
    profiles/collect.sh: separate FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE×2 gfx950 correction)."""
    p = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_force_launch")
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default=os.environ.get("MOLLYHIP_BENCH_WORKLOAD", "lj1m"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--integrator", default="vv", choices=["vv", "langevin"], help="vv = the headline VelocityVerlet step; langevin = Langevin middle integrator (single GPU)")
    ap.add_argument("--profile-steps", type=int, default=200, help="steps of the separate hipEvent-timed pass")
    ap.add_argument("--equil", type=int, default=None, help="untimed equilibration steps before the warm-up (SURVEY §8(d): 2000 for the LJ fluids, which start from a jittered lattice; 0 for 6mrr, which starts from an equilibrated structure)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N …")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    # stdout carries exactly ONE line, the JSON record: libraries that print on their own (RCCL's version banner comes out of the
    # C stdio buffer at exit, i.e. after anything printed here) are pointed at stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import molly_loader
    m = molly_loader.load()
    if m.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    case, dtype, dt = make_case(args.workload)
    if world > 1 or os.environ.get("MOLLYHIP_FORCE_DOMAIN"):   # the env hook drives the N = 1 box through the multi-GPU host loop (overhead checks)
        from molly_jl_amd import domain   # spatial decomposition + RCCL halo exchange
        result = domain.bench_distributed(m, case, dtype, dt, args, rank, local_rank, world)
        if rank != 0:
            return
        ms_per_step, st, extra = result
    else:
        import ctypes as C
        L = m.lib()
        s = case.system(m, dtype)
        s.push_state(velocities=True)
        ctx = s.engine()
        if args.integrator == "langevin":   # thermostatted at the LJ fluid's 85 K / the protein's 300 K, friction 1 / ps
            kT = m.BOLTZMANN * (85.0 if args.workload.startswith("lj") else 300.0)
            run = lambda first, n: s._check(L.mhip_langevin_run(ctx, first, n, dt, kT, 1.0, 1, 0x9E3779B97F4A7C15, first))
        else:
            run = lambda first, n: s._check(L.mhip_vv_run(ctx, first, n, dt, 1))
        # untimed setup: the LJ fluids start from a jittered lattice and are equilibrated first (SURVEY §8(d) cfg 2 / 4: "equilibrate
        # 2 000 steps before timing"), so that the timed steps see the list lifetimes of the liquid, not of a melting lattice
        equil = args.equil if args.equil is not None else (2000 if args.workload.startswith("lj") else 0)
        if equil:
            run(0, equil)
        run(equil, args.warmup)                                      # untimed warm-up
        first = equil + args.warmup
        # A window shorter than the life of an outer pair list (≈ 100 steps at 1M atoms; its search costs 2.5 ms = 17 steps) would
        # measure whether a search happens to fall into it — 0.15 or 0.30 ms/step for the same code.  Short windows are therefore
        # placed at a defined point of the list cycle: mid-cycle, 16 steps after a prune of the inner list, so that they contain the
        # next prune (every ≈ 25 steps: their fair share is 0.8) and no outer search (fair share 0.2 × 2.5 ms, in the record under
        # roofline.stage_ms_per_call).  The record says so; the 2000-step default contains twenty whole cycles.
        window = "as scheduled"
        if args.steps < 100 and equil > 0:
            def wait_for(key):
                nonlocal first
                n0 = s.stats()[key]
                for _ in range(200):
                    if s.stats()[key] != n0:
                        return True
                    run(first, 2); first += 2
                return False
            if wait_for("n_outer_builds") and wait_for("n_filter_passes"):
                run(first, 16); first += 16
                window = "mid-cycle: starts 16-18 steps after a prune of the inner pair list (contains the next prune, no outer search)"
        s._check(L.mhip_synchronize(ctx))
        t0 = time.perf_counter()
        run(first, args.steps)                                       # timed: exactly K steps; returns after a stream sync
        s._check(L.mhip_synchronize(ctx))
        ms_per_step = (time.perf_counter() - t0) * 1e3 / args.steps
        # separate pass with hipEvent stage timers on the engine's stream (never mixed into the timed region)
        s._check(L.mhip_set_profiling(ctx, 1))
        run(first + args.steps, args.profile_steps)
        st = s.stats()
        s._check(L.mhip_set_profiling(ctx, 0))
        s._check(L.mhip_check_finite(ctx))
        extra = {"timed_window": window}

    steps_s = 1e3 / ms_per_step
    ns_day = steps_s * (dt * 1e3) * 86400 * 1e-6        # dt [ps] → fs
    n_atoms = case.n
    force_ms = st["prof_ms"][0] / max(st["prof_calls"][0], 1)
    fbytes = st["force_pass_bytes"]
    achieved = fbytes / (force_ms * 1e-3) / 1e9 if force_ms > 0 else None
    roofline = {"bound": "hbm", "kernel": "k_forces", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": load_traffic(args.workload),
                "algorithmic_bytes_per_launch": fbytes, "avg_launch_ms": force_ms,
                "step_bytes": st["algorithmic_bytes_step"],
                "step_frac": st["algorithmic_bytes_step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "stage_ms_per_step": {"forces": force_ms, "build_kernel": st["prof_ms"][1] / max(args.profile_steps, 1),
                                      "list_filter": st["prof_ms"][4] / max(args.profile_steps, 1),
                                      "integrator": st["prof_ms"][2] / max(args.profile_steps, 1),
                                      "sort_permute": st["prof_ms"][3] / max(args.profile_steps, 1),
                                      "bonded": st["prof_ms"][5] / max(args.profile_steps, 1),
                                      "pme_reciprocal": st["prof_ms"][6] / max(args.profile_steps, 1)},
                "stage_ms_per_call": {"build_kernel": st["prof_ms"][1] / max(st["prof_calls"][1], 1), "list_filter": st["prof_ms"][4] / max(st["prof_calls"][4], 1)}}
    line = {
        "metric": "ns_per_day", "value": ns_day, "unit": "ns/day", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if dtype == np.float32 else "f64", "data": "synthetic",
        "matom_steps_per_s": steps_s * n_atoms / 1e6,
        "config": {"workload": {"lj1m": "1M-atom LJ fluid (argon, rho=21.1/nm3), cubic PBC, DistanceCutoff 1.0 nm, r_list 1.2 nm, dt 2 fs, VelocityVerlet, remove_CM_motion=1",
                                "lj256k": "256k-atom LJ fluid, DistanceCutoff 1.0 nm, cell-list neighbours, Float32",
                                "6mrr_pme": "6mrr (15954 atoms) Amber99SB-ILDN/TIP3P, LJ + Ewald direct space + PME reciprocal space (order 5, mesh 46x46x51, every step) + bonded + EwaldExclusion, Float32, dt 0.5 fs",
                                "6mrr_direct": "6mrr (15954 atoms) Amber99SB-ILDN/TIP3P, LJ + Ewald direct space only (no reciprocal PME) + bonded + EwaldExclusion, Float32, dt 0.5 fs",
                                "6mrr_rf64": "6mrr reaction-field Coulomb + LJ + bonded, Float64, dt 0.5 fs"}[args.workload],
                   "name": args.workload, "integrator": args.integrator, "n_atoms": n_atoms, "dt_fs": dt * 1e3, "rebuild_every": case.rebuild_every,
                   "parallelism": "single domain" if world == 1 else extra.get("parallelism"),
                   "block_atoms": st["block_atoms"], "j_split": st["j_split"], "pairs_half_list": st["n_pairs_full"] // 2,
                   "timed_window": extra.get("timed_window", "as scheduled")},
        "roofline": roofline,
        "engine": {k: st[k] for k in ("n_blocks", "block_atoms", "j_split", "max_tile_atoms", "tile_atoms_total", "lds_bytes",
                                      "n_list_slots", "n_pairs_full", "minimg_mode", "n_rebuilds", "n_outer_builds", "n_filter_passes", "last_rebuild_ms")},
    }
    line.update({k: v for k, v in extra.items() if k not in ("parallelism", "timed_window")})
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(case, dtype, dt)
    os.write(json_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()

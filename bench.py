#!/usr/bin/env python3
"""bench.py — ns/day (and Matom-steps/s) of the MI355X nonbonded engine on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W [--workload lj1m|lj256k|6mrr_pme|6mrr_direct|6mrr_rf64|6mrr_rf32|argon4096]

One "step" = one velocity-Verlet MD step (forces + integration + amortised neighbour-list upkeep) of the whole
system, state resident in HBM.  N = 1: the whole box on one GPU.  N > 1 (launched by torch.distributed.run,
one rank per GPU): the same box cut into spatial bricks with RCCL ghost-coordinate exchange — total work is
fixed, so scaling is "strong".  Rank 0 prints ONE JSON line.  At N = 1 with the default workload (the 1M-atom LJ
fluid) the line also carries, under "secondary", complete records of the metric's other configurations that fit
one GPU: 6mrr with PME (BASELINE.json configs[2]) and the 256k-atom LJ fluid (configs[1]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    "lj1m": "1M-atom LJ fluid (argon, rho=21.1/nm3), cubic PBC, DistanceCutoff 1.0 nm, r_list 1.2 nm, dt 2 fs, VelocityVerlet, remove_CM_motion=1",
    "lj256k": "256k-atom LJ fluid, DistanceCutoff 1.0 nm, cell-list neighbours, Float32",
    "6mrr_pme": "6mrr (15954 atoms) Amber99SB-ILDN/TIP3P, LJ + Ewald direct space + PME reciprocal space (order 5, mesh 46x46x51, every step) + bonded + EwaldExclusion, Float32, dt 0.5 fs",
    "6mrr_direct": "6mrr (15954 atoms) Amber99SB-ILDN/TIP3P, LJ + Ewald direct space only (no reciprocal PME) + bonded + EwaldExclusion, Float32, dt 0.5 fs",
    "6mrr_rf64": "6mrr reaction-field Coulomb + LJ + bonded, Float64, dt 0.5 fs",
    "6mrr_rf32": "6mrr reaction-field Coulomb + LJ + bonded, Float32, dt 0.5 fs, remove_CM_motion=false: the reference's benchmark/protein.jl:14-62 (nonbonded_method=:cutoff, 'CUDA f32')",
    "argon4096": "4096 argon atoms at uniformly random positions, 1400 kg/m3 (box 5.79 nm), LennardJones DistanceCutoff 1.2 nm, Float32: forces! and potential_energy call medians "
                 "with an advancing step counter — the reference's benchmark/benchmark_gpu_tiles.jl:13-165 'dense_f32' (and 'sparse_f32': the box four times as wide)",
}
FORCE_CALL_WORKLOADS = ("argon4096",)


def make_case(workload):
    """The synthetic inputs of SURVEY §8(d) (molly.jl_amd/workloads.py; the 6mrr parameter file is tests/golden/6mrr.npz)."""
    import importlib
    W = importlib.import_module("molly_jl_amd.workloads")
    if workload == "lj1m":
        return W.lj_fluid(100, seed=4, dtype=np.float32), np.float32, 0.002
    if workload == "lj256k":
        return W.lj_fluid(64, seed=2, dtype=np.float32), np.float32, 0.002
    if workload in ("6mrr_pme", "6mrr_direct", "6mrr_rf64", "6mrr_rf32"):
        dtype = np.float64 if workload == "6mrr_rf64" else np.float32
        return W.protein_6mrr("rf" if workload.startswith("6mrr_rf") else "ewald", dtype=dtype, bonded=True, pme=(workload == "6mrr_pme")), dtype, 0.0005
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(case, dtype, dt, budget_s=20.0):
    """The Molly-algorithm CPU restatement (oracle, kind "port") timed on this host's cores on a bounded
    sample of the same workload: same atoms, same parameters, a handful of steps."""
    from oracle import pyoracle as orc
    orc.build(native=True)
    cores = os.cpu_count() or 1
    nthreads = min(cores, 64)
    o = orc.from_case(case, dtype)
    o.native = True
    specific = case.bonds is not None
    general = case.pme is not None
    every = case.rebuild_every
    t0 = time.perf_counter()
    o.vv_run(1, dt, nthreads=nthreads, specific=specific, general=general)   # one step incl. the initial neighbour build + force pass
    t1 = time.perf_counter() - t0
    # steady-state sample: as many steps as fit the budget, at least one rebuild interval if affordable
    n = int(max(2, min(20, budget_s / max(t1 / 2.0, 1e-3))))
    t0 = time.perf_counter()
    o.vv_run(n, dt, first_step=1, nthreads=nthreads, specific=specific, general=general)
    t = time.perf_counter() - t0
    steps_s = n / t
    # The same algorithm on ONE core (the reference's 1-thread pair loop, src/force.jl:828-884), steady state: two runs of 1 and 3 steps
    # that start behind a rebuild step and contain none; every run begins with a neighbour search and a force pass (simulate! does),
    # so the difference of the two is two plain steps.  Their share of the rebuild every `every` steps is added from the threaded run's
    # cost model: the search is the serial part of both.
    base = ((n + 1) // every + 1) * every + 1
    t0 = time.perf_counter()
    o.vv_run(1, dt, first_step=base, nthreads=1, specific=specific, general=general)
    ta = time.perf_counter() - t0
    nb = 3 if ta < budget_s else 2
    t0 = time.perf_counter()
    o.vv_run(nb, dt, first_step=base + 1, nthreads=1, specific=specific, general=general)
    tb = time.perf_counter() - t0
    per_step = max((tb - ta) / (nb - 1), 1e-9)
    start_cost = max(ta - per_step, 0.0)                       # neighbour search + first force pass of a run
    steps_s1 = 1.0 / (per_step + max(start_cost - per_step, 0.0) / every)   # + the search's share (a search every `every` steps)
    return {"value": steps_s * dt * 1e3 * 86400 * 1e-6, "unit": "ns/day", "cores": nthreads, "kind": "port",
            "matom_steps_per_s": steps_s * case.n / 1e6,
            "sample": f"{n} velocity-Verlet steps of the full {case.n}-atom system (threaded pair loop of src/force.jl:886-969 + "
                      f"cell-list rebuild every {every} steps" + (", PME reciprocal space threaded as ewald.jl's n_threads > 1 methods (spreading on min(n, 4) private meshes, the rest over all threads)" if general else "") + f"), {nthreads} threads, -O3 -march=native",
            "one_core": {"value": steps_s1 * dt * 1e3 * 86400 * 1e-6, "unit": "ns/day", "cores": 1, "matom_steps_per_s": steps_s1 * case.n / 1e6,
                         "sample": f"steady state: ({nb}-step run − 1-step run) / {nb - 1} = {per_step:.3f} s per plain step (1-thread pair loop of src/force.jl:828-884, same system), "
                                   f"plus 1/{every} of the {max(start_cost - per_step, 0.0):.3f} s neighbour search"}}


def load_traffic(workload):
    """HBM bytes per force-kernel launch from the rocprofv3 PMC passes committed under profiles/ (collected by
    profiles/collect.sh: separate FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE×2 gfx950 correction)."""
    for tag in ("r05_", "r04_", "r03_", ""):
        p = os.path.join(ROOT, "profiles", f"{tag}traffic_{workload}.json")
        if os.path.exists(p):
            try:
                return json.load(open(p)).get("hbm_bytes_per_force_launch"), os.path.relpath(p, ROOT)
            except Exception:
                pass
    return None, None


def run_single(m, workload, args, steps, warmup, profile_steps):
    """One workload on one GPU → (record, case, dtype, dt).  Timed region: exactly `steps` steps per window, inputs resident in HBM,
    stream drained on both sides."""
    L = m.lib()
    case, dtype, dt = make_case(workload)
    s = case.system(m, dtype)
    s.push_state(velocities=True)
    ctx = s.engine()
    if args.integrator == "langevin":   # thermostatted at the LJ fluid's 85 K / the protein's 300 K, friction 1 / ps
        kT = m.BOLTZMANN * (85.0 if workload.startswith("lj") else 300.0)
        run = lambda first, n: s._check(L.mhip_langevin_run(ctx, first, n, dt, kT, 1.0, 1, 0x9E3779B97F4A7C15, first))
    else:
        run = lambda first, n: s._check(L.mhip_vv_run(ctx, first, n, dt, 0 if workload == "6mrr_rf32" else 1))   # (benchmark/protein.jl: remove_CM_motion=false)
    # untimed setup: the LJ fluids start from a jittered lattice and are equilibrated first (SURVEY §8(d) cfg 2 / 4: "equilibrate
    # 2 000 steps before timing"), so that the timed steps see the list lifetimes of the liquid, not of a melting lattice
    equil = args.equil if args.equil is not None else (2000 if workload.startswith("lj") else 0)
    if equil:
        run(0, equil)
    run(equil, warmup)                                           # untimed warm-up: exactly --warmup steps
    first = equil + warmup
    # The K timed steps are taken AS SCHEDULED — wherever searches and prunes of the pair lists happen to fall.  A window shorter than
    # the life of an outer pair list (≈ 100 steps at 1M atoms; a search costs ≈ 10 plain steps) measures 0.14 or 0.22 ms/step for the
    # same code depending on whether a search falls into it, so short windows are repeated back to back until they cover at least
    # 100 steps (one whole list cycle) and the headline is their mean; every window is K steps between two stream syncs.
    n_win = 1 if steps >= 100 else -(-100 // steps)
    win_ms = []
    for _ in range(n_win):
        s._check(L.mhip_synchronize(ctx))
        t0 = time.perf_counter()
        run(first, steps)                                        # timed: exactly K steps; returns after a stream sync
        s._check(L.mhip_synchronize(ctx))
        win_ms.append((time.perf_counter() - t0) * 1e3 / steps)
        first += steps
    ms_per_step = float(np.mean(win_ms))
    # separate pass with hipEvent stage timers on the engine's stream (never mixed into the timed region).  It follows the timed steps
    # without a pause: a mhip_get_stats in between (an export-sized kernel + host work, tens of ms) left the GPU idle long enough for
    # its clocks to drop, and a 200-step pass measured every kernel 6 % slow (force pass 92.8 against 87.1 µs on one box).
    s._check(L.mhip_set_profiling(ctx, 1))
    run(first, profile_steps)
    st = s.stats()
    s._check(L.mhip_set_profiling(ctx, 0))
    s._check(L.mhip_check_finite(ctx))
    s.close()
    extra = {"timed_window": "as scheduled" if n_win == 1 else f"as scheduled: mean of {n_win} consecutive windows of {steps} steps (one whole pair-list cycle)",
             "window_ms_per_step": {"n": n_win, "mean": ms_per_step, "min": float(min(win_ms)), "max": float(max(win_ms))},
             "list_upkeep_in_profile_pass": {"outer_searches": st["prof_calls"][1], "prunes": st["prof_calls"][4], "steps": profile_steps}}   # launches counted by the stage timers
    return ms_per_step, st, extra, case, dtype, dt


def run_force_calls(m, workload, args):
    """The reference's own GPU benchmark shape (benchmark/benchmark_gpu_tiles.jl:103-165): the median wall time of one forces! call and of
    one potential_energy call — each drained — with a step counter that advances by one per call, so that every tenth call is a rebuild step
    of the GPU neighbour finder.  --steps = the samples (the reference: 10).  One record per case (dense, sparse)."""
    import ctypes as C
    import importlib
    import torch
    W = importlib.import_module("molly_jl_amd.workloads")
    L = m.lib()
    out = []
    for name, mult, seed in (("dense_f32", 1.0, 42), ("sparse_f32", 4.0, 43)):
        case = W.argon_random(4096, mult, seed)
        s = case.system(m, np.float32)
        s.push_state(velocities=True)
        ctx = s.engine()
        f = torch.empty((case.n, 3), dtype=torch.float32, device="cuda")
        pe = C.c_double(0)
        def forces(k): s._check(L.mhip_forces(ctx, k, 0, f.data_ptr(), None, 1)); s._check(L.mhip_synchronize(ctx))
        def energy(k): s._check(L.mhip_potential_energy(ctx, k, C.byref(pe))); s._check(L.mhip_synchronize(ctx))
        for k in range(max(args.warmup, 2)):
            forces(k); energy(k)
        tf, te = [], []
        for k in range(args.steps):
            t0 = time.perf_counter(); forces(k); tf.append((time.perf_counter() - t0) * 1e3)
        for k in range(args.steps):
            t0 = time.perf_counter(); energy(k); te.append((time.perf_counter() - t0) * 1e3)
        s._check(L.mhip_set_profiling(ctx, 1))
        for k in range(args.steps, args.steps + 50):
            forces(k)
        st = s.stats()
        s._check(L.mhip_set_profiling(ctx, 0))
        kern_ms = st["prof_ms"][0] / max(st["prof_calls"][0], 1)
        rec = {"case": name, "box_nm": float(case.box[0]), "n_atoms": case.n, "forces_call_ms_median": float(np.median(tf)), "forces_call_ms_min": float(min(tf)),
               "potential_energy_call_ms_median": float(np.median(te)), "samples": args.steps,
               "pairs_half_list": st["n_pairs_full"] // 2, "k_forces_us": kern_ms * 1e3, "force_pass_bytes": st["force_pass_bytes"],
               "roofline_frac": (st["force_pass_bytes"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kern_ms > 0 else None,
               "rebuilds": st["n_rebuilds"], "block_atoms": st["block_atoms"], "j_split": st["j_split"]}
        if not args.no_cpu_baseline:      # the oracle's forces over its own list, all cores (the reference's forces! minus the neighbour search)
            from oracle import pyoracle as orc
            orc.build(native=True)
            o = orc.from_case(case, np.float32); o.native = True
            nt = min(os.cpu_count() or 1, 64)
            nl = o.neighbors("cell", nthreads=nt)
            o.forces(nl, nthreads=nt)
            t0 = time.perf_counter(); n_rep = 5
            for _ in range(n_rep):
                o.forces(nl, nthreads=nt)
            rec["cpu_forces_call_ms"] = (time.perf_counter() - t0) * 1e3 / n_rep; rec["cpu_cores"] = nt
        s.close()
        out.append(rec)
    d = out[0]
    return {"metric": "forces_call_ms_median", "value": d["forces_call_ms_median"], "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": d["forces_call_ms_median"], "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[workload], "name": workload, "n_atoms": 4096, "parallelism": "single domain", "timed_window": "median of --steps drained calls"},
            "roofline": {"bound": "hbm", "kernel": "k_forces", "achieved": d["force_pass_bytes"] / (d["k_forces_us"] * 1e-6) / 1e9 if d["k_forces_us"] else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": d["roofline_frac"], "traffic": None, "algorithmic_bytes_per_launch": d["force_pass_bytes"], "avg_launch_ms": d["k_forces_us"] * 1e-3},
            "cpu_baseline": ({"value": d["cpu_forces_call_ms"], "unit": "ms per forces call", "cores": d["cpu_cores"], "kind": "port",
                              "sample": "5 calls of the oracle's threaded pair loop over its own neighbour list, same system"} if "cpu_forces_call_ms" in d else None),
            "cases": out}


def make_record(workload, case, dtype, dt, ms_per_step, st, extra, world, args, steps, warmup, profile_steps):
    steps_s = 1e3 / ms_per_step
    ns_day = steps_s * (dt * 1e3) * 86400 * 1e-6        # dt [ps] → fs
    n_atoms = case.n
    force_ms = st["prof_ms"][0] / max(st["prof_calls"][0], 1)
    # the large one-type fluids run the plain pair pass with the integrator in its epilogue (k_forces STEP: no force array, no integrator launch).  Such a launch
    # is priced by SURVEY §8(d)'s model — every per-atom array touched once per pass — applied to what it does: the force pass's N(R_p + 3w) + 4L without the
    # force write (3w), plus the velocity read (4w) and the coordinate and velocity writes (2·4w): N(R_p + 12w) + 4L.  (The step's B_step = N(R_p + 22w) + 4L
    # stays the yardstick of step_frac below; it counts the force array's round trip, which this launch does not make.)
    fused = st.get("n_fused_steps", 0) > 0 and workload.startswith("lj")      # (6mrr's steps fuse the integrator into their LAST launch, not into the pair kernel this block is about)
    w_bytes = 4 if dtype == np.float32 else 8
    fbytes = st["force_pass_bytes"] + 9 * w_bytes * n_atoms if fused else st["force_pass_bytes"]
    achieved = fbytes / (force_ms * 1e-3) / 1e9 if force_ms > 0 else None
    traffic, traffic_src = load_traffic(workload)
    per_step = lambda k: st["prof_ms"][k] / max(profile_steps, 1)
    roofline = {"bound": "hbm", "kernel": "k_forces<STEP> (pair pass + velocity-Verlet update in its epilogue)" if fused else "k_forces", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "traffic_source": (f"{traffic_src}: rocprofv3 PMC passes of an earlier run of this command (FETCH_SIZE x2 + WRITE_SIZE), not measured in this run" if traffic_src else None),
                "algorithmic_bytes_per_launch": fbytes, "avg_launch_ms": force_ms, "fused_step": fused, "force_pass_bytes": st["force_pass_bytes"],
                "avg_launch_source": "hipEvents on the engine's stream around every plain force pass of a separate profiling pass in this run",
                "step_bytes": st["algorithmic_bytes_step"],
                "step_frac": st["algorithmic_bytes_step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "stage_ms_per_step": {"forces": per_step(0), "build_kernel": per_step(1), "list_filter": per_step(4), "integrator": per_step(2),
                                      "sort_permute": per_step(3), "bonded": per_step(5), "pme_reciprocal": per_step(6)},
                "stage_ms_per_call": {"build_kernel": st["prof_ms"][1] / max(st["prof_calls"][1], 1), "list_filter": st["prof_ms"][4] / max(st["prof_calls"][4], 1)}}
    line = {
        "metric": "ns_per_day", "value": ns_day, "unit": "ns/day", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if dtype == np.float32 else "f64", "data": "synthetic",
        "matom_steps_per_s": steps_s * n_atoms / 1e6,
        "config": {"workload": WORKLOADS[workload],
                   "name": workload, "integrator": args.integrator, "n_atoms": n_atoms, "dt_fs": dt * 1e3, "rebuild_every": case.rebuild_every,
                   "parallelism": "single domain" if world == 1 else extra.get("parallelism"),
                   "block_atoms": st["block_atoms"], "j_split": st["j_split"], "pairs_half_list": st["n_pairs_full"] // 2,
                   "timed_window": extra.get("timed_window", "as scheduled")},
        "roofline": roofline,
        "engine": {k: st[k] for k in ("n_blocks", "block_atoms", "j_split", "max_tile_atoms", "tile_atoms_total", "lds_bytes",
                                      "n_list_slots", "n_pairs_full", "minimg_mode", "n_rebuilds", "n_outer_builds", "n_filter_passes", "last_rebuild_ms")},
    }
    line.update({k: v for k, v in extra.items() if k not in ("parallelism", "timed_window")})
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default=os.environ.get("MOLLYHIP_BENCH_WORKLOAD", "lj1m"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 6mrr_pme and lj256k records that the default single-GPU run appends")
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

    if world > 1 or os.environ.get("MOLLYHIP_FORCE_DOMAIN"):   # the env hook drives the N = 1 box through the multi-GPU host loop (overhead checks)
        case, dtype, dt = make_case(args.workload)
        from molly_jl_amd import domain   # spatial decomposition + RCCL halo exchange
        result = domain.bench_distributed(m, case, dtype, dt, args, rank, local_rank, world)
        if rank != 0:
            return
        ms_per_step, st, extra = result
        line = make_record(args.workload, case, dtype, dt, ms_per_step, st, extra, world, args, args.steps, args.warmup, args.profile_steps)
    elif args.workload in FORCE_CALL_WORKLOADS:
        if args.steps > 500:
            args.steps, args.warmup = 100, 10            # (the defaults are sized for MD steps; the reference takes 10 samples)
        line = run_force_calls(m, args.workload, args)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        return
    else:
        ms_per_step, st, extra, case, dtype, dt = run_single(m, args.workload, args, args.steps, args.warmup, args.profile_steps)
        line = make_record(args.workload, case, dtype, dt, ms_per_step, st, extra, world, args, args.steps, args.warmup, args.profile_steps)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(case, dtype, dt)
        if args.workload == "lj1m" and not args.no_secondary and args.integrator == "vv":
            # the metric's other single-GPU configurations (BASELINE.json: "6mrr PME fp32 and 1M-atom LJ"; configs[1] = 256k LJ): same
            # protocol, their own step counts (both run in seconds), complete records
            line["secondary"] = []
            # (warm-up of 1 000 steps: each follows a CPU baseline leg during which the GPU sat idle and its clocks came down)
            for wl, k_steps, k_warm in (("6mrr_pme", 2000, 1000), ("lj256k", 2000, 1000)):
                ms2, st2, ex2, case2, dtype2, dt2 = run_single(m, wl, args, k_steps, k_warm, 400)
                rec = make_record(wl, case2, dtype2, dt2, ms2, st2, ex2, 1, args, k_steps, k_warm, 400)
                if not args.no_cpu_baseline:
                    rec["cpu_baseline"] = cpu_baseline(case2, dtype2, dt2, budget_s=8.0)
                line["secondary"].append(rec)
    # the LAST key of the line: every configuration's headline in a few hundred bytes, so that a reader who keeps only the tail of
    # stdout still sees all of them (round 3's driver record lost the 6mrr_pme figure inside the 14 KB line)
    recs = [line] + list(line.get("secondary", []))
    summ = {}
    for r in recs:
        nm = r["config"]["name"]
        summ[nm + "_ms_per_step"] = round(r["ms_per_step"], 5)
        summ[nm + "_ns_day"] = round(r["value"], 1)
        if r.get("roofline", {}).get("frac") is not None:
            summ[nm + "_k_forces_us"] = round(r["roofline"]["avg_launch_ms"] * 1e3, 2)
            summ[nm + "_roofline_frac"] = round(r["roofline"]["frac"], 4)
        if r.get("cpu_baseline"):
            summ[nm + "_cpu_ns_day"] = round(r["cpu_baseline"]["value"], 3)
    line["summary"] = summ
    os.write(json_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()

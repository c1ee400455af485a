#!/usr/bin/env python3
"""bench.py — ns/day (and Matom-steps/s) of the MI355X nonbonded engine on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W [--workload lj1m|lj256k|6mrr_pme|6mrr_direct|6mrr_rf64|6mrr_rf32|argon4096|memlimit]

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


def lib_build_id():
    """First 16 hex digits of the SHA-256 of the HIP library this run loads: the counter files under profiles/ carry the id of the library they
    were taken with, so a stale one is detectable."""
    import hashlib
    p = os.path.join(ROOT, "molly.jl_amd", "libmollyhip.so")
    try:
        return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


KERNEL_SOURCES = ("kernels.h", "physics.h", "common.h", "forces_launch.h", "forces_uniform.hip", "forces_inst.hip", "forces_gs.hip")


def kernel_src_id():
    """SHA-256 (16 hex digits) over the sources of the pair, search and pruning kernels (molly.jl_amd/csrc): unlike the library's hash it survives a rebuild that
    touched only the host side, so it tells whether committed PMC counters describe the kernels this run launches."""
    import hashlib
    h = hashlib.sha256()
    try:
        for f in KERNEL_SOURCES:
            h.update(open(os.path.join(ROOT, "molly.jl_amd", "csrc", f), "rb").read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def host_info():
    """the host's core count and CPU model string go into the result record."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"nproc": os.cpu_count() or 1, "usable_cores": usable, "cpu_model": model}

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
WORKLOADS["memlimit"] = ("the reference's 'Testing GPU memory limits' recipe (docs/src/examples.md:969-1017): n atoms of mass 10, sigma 0.001 nm, eps 0.1 kJ/mol at uniformly random positions, "
                         "0.013 nm3 per atom (76.9 atoms/nm3), LennardJones DistanceCutoff 1.0 nm over GPUNeighborFinder(dist_cutoff 1.0 nm), Float32, zero velocities, "
                         "VelocityVerlet dt 0.1 fs, remove_CM_motion=false, 100 steps; n doubled until a size fails, then bisected")


def make_case(workload):
    """The inputs of SURVEY §8(d) (molly.jl_amd/workloads.py): synthetic LJ fluids; 6mrr from the reference's PDB and force field (molly.jl_amd/data/6mrr_system.npz)."""
    import importlib
    W = importlib.import_module("molly_jl_amd.workloads")
    if workload == "lj1m":
        return W.lj_fluid(100, seed=4, dtype=np.float32), np.float32, 0.002
    if workload == "lj256k":
        return W.lj_fluid(64, seed=2, dtype=np.float32), np.float32, 0.002
    if workload.startswith("lj_side"):      # the benchmark fluid at any size: n_side³ atoms (lj_side50 = 125 000: what one of eight bricks of lj1m owns — DESIGN §6's scaling model)
        return W.lj_fluid(int(workload[7:]), seed=2, dtype=np.float32), np.float32, 0.002
    if workload in ("6mrr_pme", "6mrr_direct", "6mrr_rf64", "6mrr_rf32"):
        dtype = np.float64 if workload == "6mrr_rf64" else np.float32
        return W.protein_6mrr("rf" if workload.startswith("6mrr_rf") else "ewald", dtype=dtype, bonded=True, pme=(workload == "6mrr_pme")), dtype, 0.0005
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(case, dtype, dt, budget_s=20.0):
    """The Molly-algorithm CPU restatement (oracle, kind "port") timed on this host's cores on a bounded
    sample of the same workload: same atoms, same parameters, a handful of steps."""
    from oracle import pyoracle as orc
    orc.build(native=True)
    host = host_info()
    nthreads = max(1, host["usable_cores"])      # every core this process may run on (round 5 capped the leg at 64 threads)
    o = orc.from_case(case, dtype)
    o.native = True
    specific = case.bonds is not None
    general = case.pme is not None
    every = case.rebuild_every
    t0 = time.perf_counter()
    o.vv_run(1, dt, nthreads=nthreads, specific=specific, general=general)   # one step incl. the initial neighbour build + force pass
    t1 = time.perf_counter() - t0
    # The reference's threaded pair loop gives every thread a private 3N force array and sums them afterwards (force.jl:886-969): beyond some thread count
    # the sums cost more than the pairs save (1M atoms on a 256-thread host: 0.15 ns/day with 256 threads, 0.43 with 64).  The baseline is the BETTER of
    # "every usable core" and 64 threads, decided by one probe step each; both timings go into the record.
    tried = {nthreads: t1}
    if nthreads > 64:
        t0 = time.perf_counter()
        o.vv_run(1, dt, first_step=1, nthreads=64, specific=specific, general=general)
        tried[64] = time.perf_counter() - t0
        t0 = time.perf_counter()
        o.vv_run(1, dt, first_step=2, nthreads=nthreads, specific=specific, general=general)      # (the first call also paid the library's warm-up)
        tried[nthreads] = min(t1, time.perf_counter() - t0)
        if tried[64] < tried[nthreads]:
            nthreads, t1 = 64, tried[64]
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
            "threads_used": nthreads, "probe_s_per_first_step_by_threads": {str(k): v for k, v in tried.items()}, "host_nproc": host["nproc"], "host_usable_cores": host["usable_cores"], "cpu_model": host["cpu_model"],
            "matom_steps_per_s": steps_s * case.n / 1e6,
            "sample": f"{n} velocity-Verlet steps of the full {case.n}-atom system (threaded pair loop of src/force.jl:886-969 + "
                      f"cell-list rebuild every {every} steps" + (", PME reciprocal space threaded as ewald.jl's n_threads > 1 methods (spreading on min(n, 4) private meshes, the rest over all threads)" if general else "") + f"), {nthreads} threads, -O3 -march=native",
            "one_core": {"value": steps_s1 * dt * 1e3 * 86400 * 1e-6, "unit": "ns/day", "cores": 1, "matom_steps_per_s": steps_s1 * case.n / 1e6,
                         "sample": f"steady state: ({nb}-step run − 1-step run) / {nb - 1} = {per_step:.3f} s per plain step (1-thread pair loop of src/force.jl:828-884, same system), "
                                   f"plus 1/{every} of the {max(start_cost - per_step, 0.0):.3f} s neighbour search"}}


def load_traffic(workload, kind=""):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (collected by profiles/collect.sh: separate FETCH_SIZE and
    WRITE_SIZE passes, FETCH_SIZE×2 gfx950 correction).  kind "" = the plain force pass, "_prune" / "_build" = the two list-upkeep kernels.
    → (bytes, file, (hash of the library the counters were taken with, hash of its kernel sources) — None where the file predates the ids)."""
    for tag in ("r06_", "r05_", "r04_", "r03_", ""):
        p = os.path.join(ROOT, "profiles", f"{tag}traffic_{workload}{kind}.json")
        if os.path.exists(p):
            try:
                d = json.load(open(p))
                return d.get("hbm_bytes_per_force_launch", d.get("hbm_bytes_per_launch")), os.path.relpath(p, ROOT), (d.get("lib_build_id"), d.get("kernel_src_id"))
            except Exception:
                pass
    return None, None, (None, None)


def measure_traffic(workload, timeout_s=90):
    """HBM bytes per launch of the plain pass, the pruning pass and the outer search, MEASURED in this run: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they
    do not fit one pass; counters with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of this same command on a shorter schedule, each in a
    child process; FETCH_SIZE x2 (gfx950 counts 128-byte requests as 64) and KiB units as in profiles/summarize.py, whose kernel naming it shares.
    → {"plain" | "prune" | "build": {"bytes", "read", "write", "launches"}} or None (no rocprofv3, a pass failed or timed out: the record falls back to the committed files)."""
    import csv
    import glob
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or os.environ.get("MOLLYHIP_BENCH_CHILD"):
        return None
    spec = importlib.util.spec_from_file_location("mhip_summarize", os.path.join(ROOT, "profiles", "summarize.py"))
    summ = importlib.util.module_from_spec(spec); spec.loader.exec_module(summ)
    env = dict(os.environ, MOLLYHIP_BENCH_CHILD="1", TMPDIR="/tmp")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "200", "--warmup", "50", "--profile-steps", "50", "--no-cpu-baseline", "--no-secondary"]
    acc = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mhip_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd, cwd="/tmp", env=env, capture_output=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != ctr:
                        continue
                    a = acc.setdefault(summ.short(row.get("Kernel_Name", "")), {}).setdefault(ctr, [0.0, 0])
                    a[0] += float(row.get("Counter_Value", 0)); a[1] += 1
        except (subprocess.TimeoutExpired, OSError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for kind, names in (("plain", ("k_forces_step", "k_forces", "k_forces_gs")), ("prune", ("k_forces_prune",)), ("build", ("k_build",))):
        k = next((n for n in names if n in acc and "FETCH_SIZE" in acc[n] and "WRITE_SIZE" in acc[n]), None)
        if k:
            rd = acc[k]["FETCH_SIZE"][0] / max(acc[k]["FETCH_SIZE"][1], 1) * 1024 * 2
            wr = acc[k]["WRITE_SIZE"][0] / max(acc[k]["WRITE_SIZE"][1], 1) * 1024
            out[kind] = {"bytes": rd + wr, "read": rd, "write": wr, "launches": acc[k]["FETCH_SIZE"][1], "kernel": k}
    return out or None


def run_single(m, workload, args, steps, warmup, profile_steps):
    """One workload on one GPU → (record, case, dtype, dt).  Timed region: exactly `steps` steps per window, inputs resident in HBM,
    stream drained on both sides."""
    L = m.lib()
    case, dtype, dt = make_case(workload)
    s = case.system(m, dtype)
    s.push_state(velocities=True)
    ctx = s.engine()
    if args.block_atoms:      # an explicit launch shape (≙ set_cuda_launch_config!, src/cuda_config.jl:17-47) instead of the engine's choice by atom count
        s._check(L.mhip_set_launch_config(ctx, args.block_atoms, args.j_split))
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


def memlimit_trial(m, n, check=False, seed=7):
    """One size of the memory-limit recipe through the C ABI with DEVICE arrays (nothing of size n lives on the host): create → set_atoms → set_state → 100 steps →
    finiteness, pair count, stats.  → record with ok / error."""
    import ctypes as C
    import importlib
    import torch
    W = importlib.import_module("molly_jl_amd.workloads")
    _lib = importlib.import_module("molly_jl_amd._lib")
    L = m.lib()
    P = W.MEMLIMIT
    box = W.memlimit_box(n)
    rec = {"n_atoms": int(n), "box_nm": box, "ok": False}
    dev = torch.device("cuda", 0)
    ctx = C.c_void_p()
    keep = []
    try:
        torch.cuda.empty_cache()
        free0, total = torch.cuda.mem_get_info(dev)
        g = torch.Generator(device=dev); g.manual_seed(seed)
        x = torch.rand((n, 3), generator=g, device=dev, dtype=torch.float32) * box
        x = torch.where(x >= box, torch.zeros_like(x), x).contiguous()
        const = lambda v: torch.full((n,), v, device=dev, dtype=torch.float32)
        q, sig, eps, mass = const(0.0), const(P["sigma"]), const(P["eps"]), const(P["mass"])
        vel = torch.zeros((n, 3), device=dev, dtype=torch.float32)
        keep = [x, q, sig, eps, mass, vel]
        torch.cuda.synchronize()
        cfg = _lib.Config()
        cfg.precision = 32; cfg.device_id = 0; cfg.n_atoms = n
        for d in range(3):
            cfg.box[d] = box; cfg.origin[d] = 0.0; cfg.periodic[d] = 1
        cfg.rebuild_every = P["n_steps_reorder"]; cfg.r_list = P["r_cut"]
        it = _lib.Interactions()
        it.lj_weight_special = 1.0; it.coul_weight_special = 1.0; it.coul_ke = m.COULOMB_CONST; it.rf_dielectric = 1.0; it.ewald_approx_erfc = 1
        it.lj_enabled = 1; it.lj_cutoff_kind = _lib.CUTOFF_DISTANCE; it.lj_rc = P["r_cut"]
        cfg.inter = it
        rc = L.mhip_create(C.byref(ctx), C.byref(cfg))
        if rc != 0:
            raise m.MollyHipError(rc, L.mhip_last_error(None).decode())

        def chk(rc):
            if rc != 0:
                raise m.MollyHipError(rc, L.mhip_last_error(ctx).decode())
        chk(L.mhip_set_atoms(ctx, q.data_ptr(), sig.data_ptr(), eps.data_ptr(), mass.data_ptr(), None, _lib.MEM_DEVICE))
        chk(L.mhip_set_state(ctx, x.data_ptr(), vel.data_ptr(), _lib.MEM_DEVICE))
        t0 = time.perf_counter()
        chk(L.mhip_vv_run(ctx, 0, P["n_steps"], P["dt"], 0))
        chk(L.mhip_synchronize(ctx))
        rec["first_100_steps_s"] = time.perf_counter() - t0          # includes the first search and every allocation
        t0 = time.perf_counter()
        chk(L.mhip_vv_run(ctx, P["n_steps"], P["n_steps"], P["dt"], 0))
        chk(L.mhip_synchronize(ctx))
        rec["ms_per_step"] = (time.perf_counter() - t0) * 1e3 / P["n_steps"]      # the next 100 steps: 4 searches + 100 passes, everything allocated
        chk(L.mhip_check_finite(ctx))
        free1, _ = torch.cuda.mem_get_info(dev)
        rec["hbm_in_use_gb"] = (total - free1) / 1e9; rec["hbm_total_gb"] = total / 1e9
        st = _lib.Stats()
        chk(L.mhip_get_stats(ctx, C.byref(st)))
        st = st.as_dict()
        pairs = st["n_pairs_full"] // 2
        # uniformly random points in a periodic box: E[pairs] = N(N−1)/2 · (4/3)π r³ / V, variance ≈ the mean
        expect = 0.5 * n * (n - 1) * (4.0 / 3.0) * np.pi * P["r_cut"] ** 3 / (box ** 3)
        rec.update({"pairs_half_list": int(pairs), "pairs_expected": expect, "pairs_deviation_sigma": (pairs - expect) / np.sqrt(expect),
                    "list_slots": st["n_list_slots"], "block_atoms": st["block_atoms"], "j_split": st["j_split"], "n_blocks": st["n_blocks"], "max_tile_atoms": st["max_tile_atoms"],
                    "bytes_per_atom_in_hbm": (total - free1 - 0) / n, "n_rebuilds": st["n_rebuilds"]})
        f = torch.empty((n, 3), device=dev, dtype=torch.float32)
        chk(L.mhip_forces(ctx, 2 * P["n_steps"], 0, f.data_ptr(), None, _lib.MEM_DEVICE))
        chk(L.mhip_synchronize(ctx))
        fsum = f.double().sum(dim=0); fabs = f.double().abs().sum(dim=0)
        rec["net_force_over_abs_force"] = float((fsum.abs() / fabs.clamp_min(1e-300)).max())      # Newton's third law over the whole box
        rec["ok"] = bool(abs(rec["pairs_deviation_sigma"]) < 6.0)
        if not rec["ok"]:
            rec["error"] = "pair count off the closed-form expectation"
        if check and rec["ok"]:
            chk(L.mhip_get_state(ctx, x.data_ptr(), None, _lib.MEM_DEVICE))
            rec["oracle_check"] = memlimit_oracle_check(W, x, f, box)
    except m.MollyHipError as e:
        rec["error"] = str(e)[:400]
    except (RuntimeError, MemoryError) as e:      # torch's own allocations
        rec["error"] = ("torch: " + str(e).splitlines()[0])[:400]
    finally:
        if ctx:
            L.mhip_destroy(ctx)
        del keep
        torch.cuda.empty_cache()
    return rec


def memlimit_oracle_check(W, x, f, box, n_inner=100_000):
    """Forces of the ≈ 10⁵ atoms in a cube at the box centre against the fp64 oracle evaluated on that cube plus a shell of one cutoff around it (an isolated
    cluster in a box wide enough that no image interacts: every partner of an inner atom is in the cluster, so its force is the whole system's)."""
    import torch
    from oracle import pyoracle as orc
    orc.build(native=True)
    P = W.MEMLIMIT
    a = (n_inner * P["volume_per_atom"]) ** (1.0 / 3.0)
    if a + 2 * P["r_cut"] + 0.5 > box:
        return {"skipped": "box too small for an isolated cluster"}
    c = 0.5 * box
    d = (x - c).abs().amax(dim=1)
    shell = d < 0.5 * a + P["r_cut"]
    idx = torch.nonzero(shell).squeeze(1)
    xs = x[idx].double().cpu().numpy(); fs = f[idx].double().cpu().numpy()
    inner = (d[idx] < 0.5 * a).cpu().numpy()
    lo = c - 0.5 * a - P["r_cut"]
    side = a + 2 * P["r_cut"]
    nc = len(xs)
    case = W.Case(xs - lo, side + P["r_cut"] + 0.5, lj=dict(cutoff=("distance", P["r_cut"])), r_list=P["r_cut"], sigma=np.full(nc, P["sigma"]), eps=np.full(nc, P["eps"]),
                  mass=np.full(nc, P["mass"]), velocities=np.zeros((nc, 3)))
    o = orc.from_case(case, np.float64); o.native = True
    nt = host_info()["usable_cores"]
    nl = o.neighbors("cell", nthreads=nt)
    f_ref = o.forces(nl, nthreads=nt)
    scale, jump = o.force_scale(nl)
    err = np.linalg.norm(fs - f_ref, axis=1)[inner]
    tol = (4e-5 * scale + 1.01 * jump + 1e-6)[inner]
    fmag = np.linalg.norm(f_ref, axis=1)[inner]
    return {"inner_atoms": int(inner.sum()), "cluster_atoms": int(nc), "worst_err_over_tol": float((err / tol).max()), "max_abs_err": float(err.max()), "max_force": float(fmag.max()),
            "atoms_with_force_above_1e-3": int((fmag > 1e-3).sum()),
            "note": "sigma = 0.001 nm: all but the few atoms with a partner inside ~0.01 nm feel forces below 1e-6 kJ/mol/nm, so this holds the rare close pairs and the absence of "
                    "garbage; the pair count against the closed form is the check of the list", "passed": bool((err <= tol).all())}


def run_memlimit(m, args):
    """docs/src/examples.md:969-1017 on this GPU: the atom count doubled from --memlimit-start until a size fails, then four bisection steps between the last size that
    ran and the first that did not.  The reference's published results: 60 000 (RTX 2080 Ti, 11 GB), 140 000 (RTX A6000, 48 GB), 120 000 (RTX 5090, 32 GB)."""
    trials = []
    n = args.memlimit_start
    best, fail = None, None
    while True:
        r = memlimit_trial(m, n, check=(n == args.memlimit_start))
        trials.append(r)
        print(f"[memlimit] {n} atoms: {'ok' if r['ok'] else 'FAILED'} {r.get('ms_per_step', '')} ms/step, {r.get('hbm_in_use_gb', '')} GB {r.get('error', '')}", file=sys.stderr)
        if not r["ok"]:
            fail = r; break
        best = r
        if args.memlimit_max and n * 2 > args.memlimit_max:
            break
        n *= 2
    if best is not None and fail is not None:
        lo, hi = best["n_atoms"], fail["n_atoms"]
        for _ in range(4):
            mid = ((lo + hi) // 2) // 1000 * 1000
            if mid <= lo:
                break
            r = memlimit_trial(m, mid)
            trials.append(r)
            print(f"[memlimit] {mid} atoms: {'ok' if r['ok'] else 'FAILED'} {r.get('error', '')}", file=sys.stderr)
            if r["ok"]:
                lo, best = mid, r
            else:
                hi, fail = mid, r
    if best is None:
        raise SystemExit(f"memlimit: the first size failed: {trials[0].get('error')}")
    return {"metric": "max_atoms_100_steps", "value": best["n_atoms"], "unit": "atoms", "n_gpus": 1, "steps": 100, "warmup": 0, "ms_per_step": best.get("ms_per_step"),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": best["n_atoms"] / 140000.0,
            "vs_baseline_note": "the reference's largest published result for this recipe: 140 000 atoms on an NVIDIA RTX A6000 (48 GB), docs/src/examples.md:1014-1017 — other hardware, 1/6 of the memory",
            "dtype": "f32", "data": "synthetic", "lib_build_id": lib_build_id(),
            "config": {"workload": WORKLOADS["memlimit"], "name": "memlimit", "n_atoms": best["n_atoms"], "parallelism": "single domain"},
            "largest_that_ran": best, "smallest_that_failed": fail, "trials": trials,
            "reference_published": {"NVIDIA GeForce RTX 2080 Ti (11 GB)": 60000, "NVIDIA RTX A6000 (48 GB)": 140000, "NVIDIA GeForce RTX 5090 (32 GB)": 120000}}


def make_record(workload, case, dtype, dt, ms_per_step, st, extra, world, args, steps, warmup, profile_steps, live=None):
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
    traffic, traffic_src, traffic_id = load_traffic(workload)
    build_id, src_id = lib_build_id(), kernel_src_id()
    live = live or {}
    if "plain" in live:      # measured in this run (measure_traffic): the library's own ids
        traffic, traffic_src, traffic_id = live["plain"]["bytes"], None, (build_id, src_id)
    per_step = lambda k: st["prof_ms"][k] / max(profile_steps, 1)
    per_call = lambda k: st["prof_ms"][k] / max(st["prof_calls"][k], 1)

    def upkeep(kind, stage, nbytes, name):
        """roofline entry of a list-upkeep kernel: its algorithmic bytes (mhip_stats) ÷ its hipEvent-timed launch of this run; PMC traffic from its committed file"""
        ms = per_call(stage)
        if not ms or not nbytes:
            return None
        t, src, tid = load_traffic(workload, kind)
        if kind.strip("_") in live:
            t, src, tid = live[kind.strip("_")]["bytes"], "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in child processes (bench.py measure_traffic)", (build_id, src_id)
        return {"kernel": name, "avg_launch_ms": ms, "launches_in_profile_pass": st["prof_calls"][stage], "algorithmic_bytes_per_launch": nbytes,
                "achieved": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": t, "traffic_over_algorithmic": (t / nbytes) if t else None,
                "traffic_source": src, "traffic_lib_build_id": tid[0], "traffic_is_of_this_build": (tid[0] == build_id) if tid[0] else None,
                "traffic_is_of_these_kernel_sources": (tid[1] == src_id) if tid[1] else None}

    roofline = {"bound": "hbm", "kernel": "k_forces<STEP> (pair pass + velocity-Verlet update in its epilogue)" if fused else "k_forces", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "traffic_source": ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in child processes (bench.py measure_traffic), FETCH_SIZE x2 gfx950 correction" if "plain" in live else
                                   (f"{traffic_src}: rocprofv3 PMC passes of an earlier run of this command (FETCH_SIZE x2 + WRITE_SIZE), not measured in this run" if traffic_src else None)),
                "traffic_read_write": ([live["plain"]["read"], live["plain"]["write"]] if "plain" in live else None),
                "traffic_lib_build_id": traffic_id[0], "traffic_is_of_this_build": (traffic_id[0] == build_id) if traffic_id[0] else None,
                "traffic_is_of_these_kernel_sources": (traffic_id[1] == src_id) if traffic_id[1] else None,
                "traffic_over_algorithmic": (traffic / fbytes) if (traffic and fbytes) else None,
                "algorithmic_bytes_per_launch": fbytes, "avg_launch_ms": force_ms, "fused_step": fused, "force_pass_bytes": st["force_pass_bytes"],
                "avg_launch_source": "hipEvents on the engine's stream around every plain force pass of a separate profiling pass in this run",
                "step_bytes": st["algorithmic_bytes_step"],
                "step_frac": st["algorithmic_bytes_step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "stage_ms_per_step": {"forces": per_step(0), "build_kernel": per_step(1), "list_filter": per_step(4), "integrator": per_step(2),
                                      "sort_permute": per_step(3), "bonded": per_step(5), "pme_reciprocal": per_step(6)},
                "stage_ms_per_call": {"build_kernel": per_call(1), "list_filter": per_call(4)},
                # the two kernels that keep the dual pair list, next to the plain pass: stage 4 = the pruning force pass + its summary kernel,
                # stage 1 = the outer search + its summary kernel
                "list_upkeep": {"prune": upkeep("_prune", 4, st.get("prune_pass_bytes", 0), "k_forces<PRUNE> (+ k_prune_summary)"),
                                "build": upkeep("_build", 1, st.get("build_pass_bytes", 0), "k_build (+ k_build_summary)"),
                                "ms_per_step": per_step(1) + per_step(4) + per_step(3)}}
    line = {
        "metric": "ns_per_day", "value": ns_day, "unit": "ns/day", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if dtype == np.float32 else "f64",
        "data": "reference fixture (data/6mrr_equil.pdb + ff99SBildn.xml / tip3p_standard.xml through tools/param_6mrr.py; velocities data/openmm_6mrr/velocities_300K.txt)" if workload.startswith("6mrr") else "synthetic",
        "lib_build_id": build_id, "kernel_src_id": src_id,
        "matom_steps_per_s": steps_s * n_atoms / 1e6,
        "config": {"workload": WORKLOADS.get(workload, WORKLOADS["lj256k"].replace("256k-atom", f"{n_atoms}-atom") if workload.startswith("lj_side") else workload),
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
    ap.add_argument("--fail-forms", default="", help=argparse.SUPPRESS)      # tests: forms of the multi-GPU step loop ("fused", "separate launches") that the last rank gives up in the untimed part
    ap.add_argument("--traffic", default="measure", choices=["measure", "file"], help="roofline.traffic of the main workload: two rocprofv3 PMC passes of this command in child processes (default, single GPU), or the committed files under profiles/")
    ap.add_argument("--block-atoms", type=int, default=0, help="launch shape of the search and pair kernels: i-atoms per workgroup (64, 128, 256; 0 = the engine's choice)")
    ap.add_argument("--j-split", type=int, default=0, help="launch shape: waves sharing one atom's list (a power of two, block_atoms * j_split <= 1024)")
    ap.add_argument("--memlimit-start", type=int, default=1_000_000, help="--workload memlimit: first atom count (doubled until a size fails)")
    ap.add_argument("--memlimit-max", type=int, default=0, help="--workload memlimit: stop doubling beyond this atom count (0: until a size fails)")
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
    elif args.workload == "memlimit":
        line = run_memlimit(m, args)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        return
    elif args.workload in FORCE_CALL_WORKLOADS:
        if args.steps > 500:
            args.steps, args.warmup = 100, 10            # (the defaults are sized for MD steps; the reference takes 10 samples)
        line = run_force_calls(m, args.workload, args)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        return
    else:
        ms_per_step, st, extra, case, dtype, dt = run_single(m, args.workload, args, args.steps, args.warmup, args.profile_steps)
        # the counters behind roofline.traffic, taken in THIS run (two PMC passes of the same command in child processes, ≈ 20 s each) unless switched off; a run that
        # is itself such a child, or a box without rocprofv3, reads the committed files and says so
        live = measure_traffic(args.workload) if args.traffic == "measure" and not args.block_atoms else None
        line = make_record(args.workload, case, dtype, dt, ms_per_step, st, extra, world, args, args.steps, args.warmup, args.profile_steps, live=live)
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
            if r["roofline"].get("step_frac") is not None:
                summ[nm + "_step_frac"] = round(r["roofline"]["step_frac"], 4)
            if r["roofline"].get("traffic_over_algorithmic") is not None:
                summ[nm + "_traffic_over_algorithmic"] = round(r["roofline"]["traffic_over_algorithmic"], 3)
            up = r["roofline"].get("list_upkeep") or {}
            if up.get("ms_per_step") is not None and nm.startswith("lj"):
                summ[nm + "_list_upkeep_ms_per_step"] = round(up["ms_per_step"], 5)
                for k in ("prune", "build"):
                    if up.get(k):
                        summ[f"{nm}_{k}_roofline_frac"] = round(up[k]["frac"], 4)
        if r.get("cpu_baseline"):
            summ[nm + "_cpu_ns_day"] = round(r["cpu_baseline"]["value"], 3)
    line["summary"] = summ
    os.write(json_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()

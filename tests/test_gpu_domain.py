"""The HIP side of the multi-GPU path on ONE GPU: 2 / 4 / 8 processes share cuda:0, each drives its own libmollyhip
context over an open (non-periodic) sub-domain with ghost atoms; collectives go over gloo with host staging (RCCL refuses
several ranks on one device).  Checks the ghost gather/scatter kernels, open-axis cell grids, owned/ghost sorting and the
device-side remove_CM reduction against the single-domain oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import systems as S

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _case(n_side, dtype, shift=0.0, temperature=85.0):
    """the 4 096-atom fluid of this file.  shift (nm, all axes): the lattice planes of the stock system lie half a spacing from every brick face, so
    nobody changes owner within a short run; shifted by 0.17 nm a plane sits 0.01 nm under each face and atoms cross it from the first steps on"""
    case = S.lj_fluid(n_side, dtype=dtype, rebuild_every=10, temperature=temperature)
    if shift:
        box = case.box[0]
        x = case.coords + shift
        x = x - np.floor(x / box) * box
        x = x.astype(dtype).astype(np.float64)
        case.coords = np.where(x >= box, 0.0, x)
    return case


def _worker(rank, world, port, n_side, n_steps, dtype_name, out_dir, gm=0.0, host_skin=None, chunk=0, tag="result", shift=0.0, temperature=85.0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import molly_loader
    molly_loader.load()
    from molly_jl_amd import domain
    dtype = np.float32 if dtype_name == "f32" else np.float64
    tdtype = torch.float32 if dtype_name == "f32" else torch.float64
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    case = _case(n_side, dtype, shift, temperature)
    grid = domain.choose_grid(world, case.box)
    bg = domain.BrickGrid(case.box, grid, rank, case.r_list + gm)
    box, origin, periodic = bg.engine_box(pad=0.3)
    eng = domain.HipDomainEngine(domain.make_interactions(case, dtype), dtype, case.n, box, origin, periodic, case.r_list, case.rebuild_every, 0, ghost_margin=gm)
    run = domain.DomainRun(bg, eng, tdtype, dev, case.rebuild_every, ghost_margin=gm, skin=(case.r_list - 1.0) if host_skin is None else host_skin)
    run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
    if chunk > 0:      # the same steps in consecutive calls (chunk boundaries on multiples of the rebuild cadence when chunk % 10 == 0)
        for first in range(0, n_steps, chunk):
            run.run(first, min(chunk, n_steps - first), 0.002, remove_cm_every=1)
    else:
        run.run(0, n_steps, 0.002, remove_cm_every=1)
    xs, vs = run.gather_global(case.n)
    if rank == 0:
        st = eng.stats()
        np.savez(os.path.join(out_dir, tag + ".npz"), engine_loop=int(run.engine_loop), x=xs, v=vs, ghosts=run.n_ghost, migrated=run.stats["migrated"], plans=run.stats["plans"],
                 dev_replans=eng.domain_info()[2] if getattr(run, "device_replan", False) else 0,
                 checks=run.stats["plan_checks"], outer=st["n_outer_builds"], prunes=st["n_filter_passes"], host_prunes=run.stats["prunes"], fused=int(run.fused), fused_steps=st["n_fused_steps"])
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype_name", [(2, "f64"), (4, "f64"), (8, "f64"), (2, "f32")])
def test_hip_domains_match_single_domain_oracle(world, dtype_name, tmp_path):
    n_side, n_steps = 16, 25          # 4096 atoms, box 5.79 nm → bricks 2.9 nm; rebuild + migration at steps 10, 20
    mp.spawn(_worker, args=(world, _free_port(), n_side, n_steps, dtype_name, str(tmp_path)), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    dtype = np.float32 if dtype_name == "f32" else np.float64
    case = S.lj_fluid(n_side, dtype=dtype, rebuild_every=10)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    if dtype_name == "f64":
        assert np.abs(d).max() < 1e-9 and np.abs(res["v"] - o.vel).max() < 1e-8
    else:
        assert np.abs(d).mean() < 1e-5 and np.abs(d).max() < 1e-3
    assert int(res["ghosts"]) > 0


@pytest.mark.parametrize("world,dtype_name,gm,n_steps,skin_pm,sched", [(1, "f64", 0.2, 60, None, "engine"), (2, "f64", 0.2, 60, None, "engine"), (8, "f64", 0.2, 40, None, "engine"), (4, "f32", 0.2, 40, None, "engine"),
                                                                          (2, "f64", 0.2, 60, 30, "engine"), (2, "f64", 0.2, 60, 30, "host"), (2, "f64", 0.03, 60, 20, "host")])
def test_long_lived_ghost_plans_with_dual_list(world, dtype_name, gm, n_steps, skin_pm, sched, tmp_path, monkeypatch):
    """ghost shell r_list + margin: ownership, ghost set and the engine's outer pair list live until a prune comes due after an
    atom moved margin/2; the prunes (dual pair list with ghosts) happen on every rank at the same step.  sched "engine": decided by
    mhip_plan_decide from the all-reduced displacements with the engine's inner skin (skin_pm fixes it, in pm, to force prunes within
    a short run); "host": by the host against the skin it is given (mhip_plan_disp2_dev + mhip_request_prune; the engine's real skin
    is 0.2 nm)."""
    n_side = 16                       # bricks 2.9 nm >= 1.2 + 0.2
    host_skin = None
    if sched == "host":
        monkeypatch.setenv("MOLLYHIP_HOST_PRUNE", "1"); host_skin = skin_pm * 1e-3
    elif skin_pm is not None:
        monkeypatch.setenv("MOLLYHIP_INNER_SKIN_PM", str(skin_pm)); monkeypatch.setenv("MOLLYHIP_INNER_SKIN_FIXED", "1")
    mp.spawn(_worker, args=(world, _free_port(), n_side, n_steps, dtype_name, str(tmp_path), gm, host_skin), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    dtype = np.float32 if dtype_name == "f32" else np.float64
    case = S.lj_fluid(n_side, dtype=dtype, rebuild_every=10)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    if dtype_name == "f64":
        assert np.abs(d).max() < 1e-9 and np.abs(res["v"] - o.vel).max() < 1e-8
    else:
        assert np.abs(d).mean() < 1e-5 and np.abs(d).max() < 2e-3
    assert int(res["checks"]) >= n_steps // 10 and int(res["fused"]) == 1      # (plus the extra checks the engine asks for when a list is good for a few more steps only)
    if dtype_name == "f32" and sched == "engine" and gm >= 0.2:
        # the fp32 one-type fluid takes the plain steps of a long-lived plan as ONE launch each (kernels.h HaloStep: wait + ghosts from the receive half in the
        # prologue, integrator + peer stores in the epilogue); prune steps and the last step of a call keep the separate launches
        assert int(res["fused_steps"]) >= n_steps // 2, int(res["fused_steps"])
    if gm >= 0.2:
        assert int(res["plans"]) == 1 and int(res["outer"]) == 1                                      # one plan, one search
        assert int(res["prunes"]) >= 1 and int(res["host_prunes"]) <= int(res["prunes"]) <= 1 + int(res["host_prunes"])   # (a prune requested after the last step is never run)
        if skin_pm is not None:
            assert int(res["host_prunes"]) >= 2
    else:
        assert int(res["plans"]) > 1 and int(res["host_prunes"]) >= 1                                 # a due prune finds the plan stale


def _run_variant(tmp_path, monkeypatch, tag, world, n_steps, engine_loop, chunk, gm=0.2, skin_pm=30, shift=0.0, temperature=85.0, dtype_name="f64"):
    monkeypatch.setenv("MOLLYHIP_ENGINE_LOOP", "1" if engine_loop else "0")
    if skin_pm is not None:
        monkeypatch.setenv("MOLLYHIP_INNER_SKIN_PM", str(skin_pm)); monkeypatch.setenv("MOLLYHIP_INNER_SKIN_FIXED", "1")
    else:
        monkeypatch.delenv("MOLLYHIP_INNER_SKIN_PM", raising=False); monkeypatch.delenv("MOLLYHIP_INNER_SKIN_FIXED", raising=False)
    mp.spawn(_worker, args=(world, _free_port(), 16, n_steps, dtype_name, str(tmp_path), gm, None, chunk, tag, shift, temperature), nprocs=world, join=True)
    return np.load(os.path.join(tmp_path, tag + ".npz"))


@pytest.mark.parametrize("world", [2, 4])
def test_engine_loop_chunked_on_cadence_boundaries_keeps_every_check(world, tmp_path, monkeypatch):
    """Two / four ranks on the one GPU, the in-engine exchange (peer stores into IPC-mapped regions, mhip_domain_run).  A run cut into
    calls whose boundaries fall on multiples of the rebuild cadence must take the same validity checks, arrange the same prunes and
    end in the same state as the run in one call — a check issued at the last step of a call is read by the first step of the next
    one, not dropped (the inner list then went unvouched for up to 2·every steps) — and both must agree with the host
    loop (MOLLYHIP_ENGINE_LOOP=0: all_to_all + the synchronous collective decision)."""
    n_steps = 60
    whole = _run_variant(tmp_path, monkeypatch, "whole", world, n_steps, True, 0)
    cut = _run_variant(tmp_path, monkeypatch, "cut", world, n_steps, True, 10)
    cut7 = _run_variant(tmp_path, monkeypatch, "cut7", world, n_steps, True, 7)        # boundaries off the cadence
    host = _run_variant(tmp_path, monkeypatch, "host", world, n_steps, False, 0)
    assert int(whole["engine_loop"]) == 1 and int(cut["engine_loop"]) == 1 and int(host["engine_loop"]) == 0
    for other in (cut, cut7):
        assert int(other["checks"]) == int(whole["checks"]) and int(other["host_prunes"]) == int(whole["host_prunes"])
        assert int(other["prunes"]) == int(whole["prunes"]) and int(other["plans"]) == int(whole["plans"]) and int(other["outer"]) == int(whole["outer"])
        assert np.abs(other["x"] - whole["x"]).max() < 1e-11 and np.abs(other["v"] - whole["v"]).max() < 1e-10
    assert int(whole["checks"]) >= n_steps // 10 and int(whole["host_prunes"]) >= 2
    # the host loop decides at the check step itself, the engine loop one step later: the same physics within fp64 round-off of
    # differently ordered sums, and the lists are pruned about as often
    d = whole["x"] - host["x"]
    assert np.abs(d).max() < 1e-9 and np.abs(whole["v"] - host["v"]).max() < 1e-8
    assert abs(int(whole["host_prunes"]) - int(host["host_prunes"])) <= 1 and int(whole["plans"]) == int(host["plans"])


@pytest.mark.parametrize("world,dtype_name", [(2, "f32"), (4, "f64")])
def test_prune_summary_read_late_is_the_same_run(world, dtype_name, tmp_path, monkeypatch):
    """Inside mhip_domain_run a pruning pass on a ghosted sub-domain no longer drains the stream for its summary (largest pruned tile, row total, displacement since the
    outer search): the words arrive behind an event and the next passes are shaped for the outer list's largest tile until they are read (engine.hip, prune_resolve).
    MOLLYHIP_PRUNE_LATE=0 keeps the drain.  Both forms must prune equally often and end in the SAME state: the tile bound changes the LDS carve-up, never a sum."""
    n_steps = 60
    monkeypatch.setenv("MOLLYHIP_PRUNE_LATE", "1")
    late = _run_variant(tmp_path, monkeypatch, "late", world, n_steps, True, 0, dtype_name=dtype_name)
    monkeypatch.setenv("MOLLYHIP_PRUNE_LATE", "0")
    drained = _run_variant(tmp_path, monkeypatch, "drained", world, n_steps, True, 0, dtype_name=dtype_name)
    monkeypatch.delenv("MOLLYHIP_PRUNE_LATE")
    assert int(late["engine_loop"]) == 1 and int(late["prunes"]) >= 2 and int(late["prunes"]) == int(drained["prunes"]) and int(late["outer"]) == int(drained["outer"])
    assert np.array_equal(late["x"], drained["x"]) and np.array_equal(late["v"], drained["v"])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_fused_ghosted_step_with_the_blocks_own_waits(world, tmp_path, monkeypatch):
    """On separate devices nothing but the bounded waits INSIDE the fused ghosted launch orders a step against its peers: blocks whose tile holds ghosts wait for the
    senders' sequence words, the head workgroup for the centre-of-mass rows, sending blocks without ghosts for the peer's read of the half they overwrite (kernels.h
    HaloStep).  Ranks that share a device normally get a one-workgroup waiter in front, which makes every one of those waits return at once; MOLLYHIP_HALO_WAITER=0 takes it
    away (safe here: two, four or eight grids of at most 32 workgroups fit the device side by side; with eight ranks every rank waits for seven peers), so that the waits really order the ranks.  Same bits as with the waiter."""
    n_steps = 60
    monkeypatch.setenv("MOLLYHIP_HALO_WAITER", "0")
    bare = _run_variant(tmp_path, monkeypatch, "bare", world, n_steps, True, 0, dtype_name="f32")
    monkeypatch.setenv("MOLLYHIP_HALO_WAITER", "1")
    waiter = _run_variant(tmp_path, monkeypatch, "waiter", world, n_steps, True, 0, dtype_name="f32")
    monkeypatch.delenv("MOLLYHIP_HALO_WAITER")
    assert int(bare["engine_loop"]) == 1 and int(bare["fused_steps"]) >= n_steps // 2 and int(bare["fused_steps"]) == int(waiter["fused_steps"])
    assert np.array_equal(bare["x"], waiter["x"]) and np.array_equal(bare["v"], waiter["v"])


@pytest.mark.parametrize("gm,n_steps", [(0.0, 40), (0.03, 120)])
def test_engine_loop_replans_match_host_loop(gm, n_steps, tmp_path, monkeypatch):
    """Re-plans (migration + new ghost routes) BETWEEN engine calls.  No ghost margin: ownership and ghosts are redone at every rebuild
    step, so mhip_domain_run returns every 10 steps; a thin margin (0.03 nm): the plan goes stale when somebody has moved half of it.
    The engine loop and the host loop must re-plan equally often and end in the same state."""
    eng = _run_variant(tmp_path, monkeypatch, "eng", 2, n_steps, True, 20, gm=gm, skin_pm=20)
    host = _run_variant(tmp_path, monkeypatch, "host", 2, n_steps, False, 20, gm=gm, skin_pm=20)
    assert int(eng["engine_loop"]) == 1 and int(host["engine_loop"]) == 0
    if gm == 0.0:
        assert int(eng["plans"]) >= n_steps // 10
    assert abs(int(eng["plans"]) - int(host["plans"])) <= 1          # (the engine loop reads its collective check one step late)
    assert np.abs(eng["x"] - host["x"]).max() < 1e-9 and np.abs(eng["v"] - host["v"]).max() < 1e-8


@pytest.mark.parametrize("world,gm,n_steps,temperature,skin_pm", [(2, 0.0, 40, 85.0, 20), (4, 0.0, 40, 85.0, 20), (8, 0.0, 30, 85.0, 20), (2, 0.05, 300, 300.0, None), (4, 0.05, 300, 300.0, None)])
def test_device_replan_matches_host_planner_and_oracle(world, gm, n_steps, temperature, skin_pm, tmp_path, monkeypatch):
    """The re-plan inside the engine (mhip_set_domain; replan.h): migration, ghost selection and the per-step message tables made by device
    compactions and peer stores in the middle of mhip_domain_run's step.  Against the host planner driving the same engine loop
    (MOLLYHIP_DEVICE_REPLAN=0: mhip_domain_run returns for every re-plan) and against the single-domain oracle: the same re-plans, atoms
    that really changed owner (the lattice is shifted so that a plane of atoms sits 0.01 nm under every brick face), the same trajectory to
    fp64 round-off of differently ordered sums.  No margin: a re-plan at every rebuild step; a thin margin on a hot fluid (300 K: it melts
    within the run): re-plans when the collective check finds the plan stale, prunes of the dual list with ghosts in between."""
    kw = dict(gm=gm, skin_pm=skin_pm, shift=0.17, temperature=temperature)
    monkeypatch.setenv("MOLLYHIP_DEVICE_REPLAN", "1")
    dev = _run_variant(tmp_path, monkeypatch, "dev", world, n_steps, True, 0, **kw)
    monkeypatch.setenv("MOLLYHIP_DEVICE_REPLAN", "0")
    host = _run_variant(tmp_path, monkeypatch, "hostplan", world, n_steps, True, 0, **kw)
    assert int(dev["engine_loop"]) == 1 and int(host["engine_loop"]) == 1
    assert int(dev["dev_replans"]) >= (n_steps // 10 if gm == 0.0 else 1) and int(host["dev_replans"]) == 0, (int(dev["dev_replans"]), int(dev["plans"]), int(host["plans"]))
    slack = 0 if gm == 0.0 else 1      # (a stale plan is found by the same check in both; the searches behind it see the step from different sides)
    assert abs(int(dev["plans"]) - int(host["plans"])) <= slack and abs(int(dev["outer"]) - int(host["outer"])) <= slack
    assert int(dev["migrated"]) > 0 and int(host["migrated"]) > 0
    assert np.abs(dev["x"] - host["x"]).max() < 1e-9 and np.abs(dev["v"] - host["v"]).max() < 1e-8
    case = _case(16, np.float64, 0.17, temperature)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = dev["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-9 and np.abs(dev["v"] - o.vel).max() < 1e-8


def test_device_replan_fp32_and_chunked_run_is_the_same_run(tmp_path, monkeypatch):
    """a run cut into calls (off the rebuild cadence) re-plans at the same steps and ends in the same state as the run in one call; the fp32 engine
    re-plans like the fp64 one (its records travel as 32-bit words, the global ids as two of them) and stays inside the fp32 trajectory bar"""
    kw = dict(gm=0.0, skin_pm=20, shift=0.17)
    whole = _run_variant(tmp_path, monkeypatch, "whole", 2, 60, True, 0, **kw)
    cut = _run_variant(tmp_path, monkeypatch, "cut", 2, 60, True, 7, **kw)
    assert int(whole["dev_replans"]) >= 6 and int(cut["dev_replans"]) == int(whole["dev_replans"]) and int(whole["migrated"]) > 0
    assert np.abs(cut["x"] - whole["x"]).max() < 1e-11 and np.abs(cut["v"] - whole["v"]).max() < 1e-10
    f32 = _run_variant(tmp_path, monkeypatch, "f32", 4, 40, True, 0, dtype_name="f32", **kw)
    assert int(f32["dev_replans"]) >= 4 and int(f32["migrated"]) > 0
    case = _case(16, np.float32, 0.17)
    o = case.oracle(np.float64)
    o.vv_run(40, 0.002, remove_cm_every=1)
    d = f32["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).mean() < 1e-5 and np.abs(d).max() < 2e-3


def _big_worker(rank, world, port, n_side, n_steps, out_dir, gm, with_pairs, shift):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import molly_loader
    molly_loader.load()
    from molly_jl_amd import domain
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    case = _case(n_side, np.float32, shift)
    grid = domain.choose_grid(world, case.box)
    bg = domain.BrickGrid(case.box, grid, rank, case.r_list + gm)
    box, origin, periodic = bg.engine_box(pad=0.3)
    vol_frac = np.prod([b / L for b, L in zip(box, case.box)])
    capacity = int(case.n * min(1.0, vol_frac) * 1.25) + 4096            # (bench_distributed's sizing)
    eng = domain.HipDomainEngine(domain.make_interactions(case, np.float32), np.float32, capacity, box, origin, periodic, case.r_list, case.rebuild_every, 0, ghost_margin=gm)
    run = domain.DomainRun(bg, eng, torch.float32, dev, case.rebuild_every, ghost_margin=gm, skin=case.r_list - 1.0)
    run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
    run.run(0, n_steps, 0.002, remove_cm_every=1)
    xs, vs = run.gather_global(case.n)
    extra = {}
    if with_pairs:      # this rank's list of NOW in local indices, its local coordinates (owned then ghosts) and the owned atoms' global ids
        i, j = eng.export_neighbors()
        x_all = torch.empty((run.n_owned + run.n_ghost, 3), dtype=torch.float32, device=dev)
        eng.get_state(x_all, run.v)
        extra = dict(pi=i, pj=j, x_all=x_all.cpu().numpy(), gid=run.gid.cpu().numpy(), n_owned=run.n_owned)
    st = eng.stats()
    np.savez(os.path.join(out_dir, f"big{rank}.npz"), x=xs, v=vs, engine_loop=int(run.engine_loop), dev_replans=eng.domain_info()[2] if run.device_replan else 0,
             migrated=run.stats["migrated"], plans=run.stats["plans"], ghosts=run.n_ghost, outer=st["n_outer_builds"], prunes=st["n_filter_passes"], block_atoms=st["block_atoms"],
             fused_steps=st["n_fused_steps"], **extra)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,gm,n_steps,with_pairs", [(2, 0.0, 20, True), (8, 0.0, 20, True), (8, 0.2, 25, False)])
def test_benchmark_size_bricks_against_single_domain_and_oracle_list(pkg, world, gm, n_steps, with_pairs, tmp_path):
    """BASELINE.json configs[1]'s fluid (262 144 atoms, fp32) cut 2×1×1 and 2×2×2 on the one GPU, with the benchmark's own capacity rule: tile
    counts, row capacities and 30 000+ ghosts per rank that the 4 096-atom cases never reach.  Steps across rebuilds and migrations (no margin:
    a re-plan inside the engine at steps 10 and 20, the lattice shifted so that atoms change owner; the benchmark's 0.2 nm: one plan, prunes only).
    Coordinates against the single-domain engine at the fp32 trajectory bar of the small cases; and, where the run ends on a re-plan step, the
    UNION of the ranks' exported neighbour lists — ghosts identified by their coordinates — against the fp32 oracle's list of the gathered
    coordinates (neighbors.jl:409-411): the same pairs, up to those that sit within rounding of r_list (a ghost's coordinate is the owner's
    plus a box length, rounded once more)."""
    n_side, shift = 64, 0.17
    mp.spawn(_big_worker, args=(world, _free_port(), n_side, n_steps, str(tmp_path), gm, with_pairs, shift), nprocs=world, join=True)
    res = [np.load(os.path.join(tmp_path, f"big{r}.npz")) for r in range(world)]
    case = _case(n_side, np.float32, shift)
    s = case.system(pkg, np.float32)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), n_steps)
    xs = res[0]["x"]
    d = xs - s.coords.astype(np.float64)
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).mean() < 1e-5 and np.abs(d).max() < 1e-3, (np.abs(d).mean(), np.abs(d).max())
    assert all(int(r["engine_loop"]) == 1 and int(r["ghosts"]) > 10000 for r in res)
    if gm < 0.2:
        assert all(int(r["dev_replans"]) >= 2 for r in res) and sum(int(r["migrated"]) for r in res) > 100, [int(r["migrated"]) for r in res]
    else:
        assert all(int(r["plans"]) == 1 and int(r["outer"]) == 1 and int(r["prunes"]) >= 1 for r in res)
        assert all(int(r["fused_steps"]) >= n_steps // 2 for r in res), [int(r["fused_steps"]) for r in res]      # the ghosted step as ONE launch on every rank
    if not with_pairs:
        return
    from scipy.spatial import cKDTree
    box = float(case.box[0])
    xw = xs - np.floor(xs / box) * box
    xw[xw >= box] = 0.0
    tree = cKDTree(xw, boxsize=box)
    keys = []
    for r in res:
        n_owned, gid, x_all = int(r["n_owned"]), r["gid"], r["x_all"].astype(np.float64)
        gw = x_all[n_owned:] - np.floor(x_all[n_owned:] / box) * box
        gw[gw >= box] = 0.0
        dist_g, idx_g = tree.query(gw, k=1)
        assert dist_g.max() < 1e-4                                            # every ghost IS some owner's atom (to the rounding of the shift)
        lg = np.concatenate([gid, idx_g.astype(np.int64)])
        assert np.array_equal(np.sort(np.unique(gid)), np.sort(gid)) and np.abs(x_all[:n_owned] - xs[gid]).max() < 1e-6
        keys.append(S.pair_keys(lg[r["pi"]], lg[r["pj"]]))
    got = np.unique(np.concatenate(keys))
    o32 = case.oracle(np.float32, coords=xs)
    oi, oj, _ = o32.neighbors("cell", nthreads=16)
    ref = S.pair_keys(oi, oj)
    only_got, only_ref = np.setdiff1d(got, ref, assume_unique=True), np.setdiff1d(ref, got, assume_unique=True)
    assert len(only_got) + len(only_ref) <= 1e-6 * len(ref), (len(only_got), len(only_ref), len(ref))
    for k in np.concatenate([only_got, only_ref]):                            # … and those few sit on the list radius
        a, b = int(k >> np.uint64(32)), int(k & np.uint64(0xffffffff))
        dv = xs[a] - xs[b]; dv -= np.round(dv / box) * box
        assert abs(np.linalg.norm(dv) - case.r_list) < 1e-5

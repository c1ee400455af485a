"""world_size 2, 4 and 8 runs of the spatial decomposition (molly_jl_amd.domain) over gloo on CPU: brick ownership, ghost
plan, per-step all_to_all ghost exchange, remove_CM all-reduce and migration, checked against the single-domain oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import systems as S


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_side, n_steps, out_dir, gm=0.0, skin=0.2, msg_dtype=torch.float64):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import molly_loader
    molly_loader.load()
    from molly_jl_amd import domain
    from tests.oracle_domain_engine import OracleDomainEngine
    case = S.lj_fluid(n_side, dtype=np.float64, rebuild_every=5)
    grid = domain.choose_grid(world, case.box)
    bg = domain.BrickGrid(case.box, grid, rank, case.r_list + gm)
    box, origin, periodic = bg.engine_box(pad=0.3)
    eng = OracleDomainEngine(case.inter_dict(np.float64), case.box, periodic, case.r_list, ghost_margin=gm, skin=skin, every=case.rebuild_every)
    run = domain.DomainRun(bg, eng, msg_dtype, torch.device("cpu"), case.rebuild_every, ghost_margin=gm, skin=skin)
    run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
    # every atom is owned exactly once
    n_tot = torch.tensor([run.n_owned]); dist.all_reduce(n_tot)
    assert int(n_tot) == case.n
    run.run(0, n_steps, 0.002, remove_cm_every=1)
    xs, vs = run.gather_global(case.n)
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), x=xs, v=vs, ghosts=run.n_ghost, migrated=run.stats["migrated"], grid=np.array(grid),
                 plans=run.stats["plans"], checks=run.stats["plan_checks"], prunes=run.stats["prunes"], fused=int(run.fused),
                 decide_steps=np.array(eng.decide_steps, dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_decomposed_run_matches_single_domain_oracle(world, tmp_path):
    n_side, n_steps = 10, 12      # 1000 atoms, box 3.6 nm → bricks 1.8 nm ≥ r_list 1.2 nm; rebuild + migration at steps 5, 10
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_side, n_steps, str(tmp_path)), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    case = S.lj_fluid(n_side, dtype=np.float64, rebuild_every=5)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-9 and np.abs(res["v"] - o.vel).max() < 1e-8
    assert int(res["ghosts"]) > 0
    assert tuple(res["grid"]) == {2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}[world]      # 8 ranks: every axis cut, seven peers each


def test_float32_messages_carry_the_momentum_sums_exactly(tmp_path):
    """fp32 runs: the ghost message is float32, the four doubles of Σ m v ride in three rows of it as raw words (mhip_halo_plan,
    cm_rows = 3).  The stand-in engine computes in float64, so coordinates lose precision on the wire, the momentum sums must not:
    with remove_CM_motion on, the total momentum after the run is zero to double rounding."""
    world, n_side, n_steps = 2, 10, 12
    mp.spawn(_worker, args=(world, _free_port(), n_side, n_steps, str(tmp_path), 0.3, 0.2, torch.float32), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    case = S.lj_fluid(n_side, dtype=np.float64, rebuild_every=5)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 2e-5 and int(res["fused"]) == 1          # float32 ghost coordinates: 1e-7 relative per exchange
    mv = res["v"] * case.mass[:, None]
    assert np.abs(mv.sum(axis=0)).max() < 1e-6 * np.abs(mv).sum(axis=0).max()   # (v comes back through float32 tensors: 1e-7 per atom, random signs)


def test_halo_layout_of_a_ghost_plan(tmp_path):
    """the index lists of mhip_halo_plan built by DomainRun._halo_layout: each peer's rows followed by cm_rows momentum rows, on both sides"""
    world = 2
    mp.spawn(_layout_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = np.load(os.path.join(tmp_path, "layout.npz"))
    n_send, n_recv, cr = int(r["n_send"]), int(r["n_ghost"]), int(r["cm_rows"])
    idx, dst, pos = r["send_idx"], r["recv_dst"], r["cm_pos"]
    assert cr == 2 and len(idx) == n_send + cr and len(dst) == n_recv + cr          # one peer: its rows, then the momentum rows
    assert (idx[:n_send] >= 0).all() and list(idx[n_send:]) == [-1, -2] and list(pos) == [n_send, n_send + 1]
    assert list(dst[:n_recv]) == list(range(n_recv)) and list(dst[n_recv:]) == [-1, -2]
    assert list(r["sc3"]) == [0, 3 * (n_send + cr)] and list(r["rc3"]) == [0, 3 * (n_recv + cr)]


def _layout_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import molly_loader
    molly_loader.load()
    from molly_jl_amd import domain
    from tests.oracle_domain_engine import OracleDomainEngine
    case = S.lj_fluid(10, dtype=np.float64, rebuild_every=5)
    grid = domain.choose_grid(world, case.box)
    bg = domain.BrickGrid(case.box, grid, rank, case.r_list + 0.3)
    box, origin, periodic = bg.engine_box(pad=0.3)
    eng = OracleDomainEngine(case.inter_dict(np.float64), case.box, periodic, case.r_list, ghost_margin=0.3, skin=0.2, every=5)
    run = domain.DomainRun(bg, eng, torch.float64, torch.device("cpu"), 5, ghost_margin=0.3, skin=0.2)
    run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
    if rank == 0:
        np.savez(os.path.join(out_dir, "layout.npz"), n_send=run.send_idx.numel(), n_ghost=run.n_ghost, cm_rows=run.cm_rows, send_idx=run.f_send_idx.numpy(),
                 recv_dst=run.f_recv_dst.numpy(), cm_pos=run.f_cm_pos.numpy(), sc3=np.array(run._fsc3), rc3=np.array(run._frc3))
    dist.barrier()
    dist.destroy_process_group()


def test_stepwise_host_loop_matches_single_domain_oracle(tmp_path, monkeypatch):
    """the two-call form of the step (halo_begin / halo_end, Σ m v all-reduced every step, prunes scheduled by the host): what grids
    with more than two bricks per axis fall back to"""
    monkeypatch.setenv("MOLLYHIP_HALO_FUSED", "0"); monkeypatch.setenv("MOLLYHIP_HOST_PRUNE", "1")
    world, n_side, n_steps = 2, 10, 12
    mp.spawn(_worker, args=(world, _free_port(), n_side, n_steps, str(tmp_path), 0.3, 0.012), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    case = S.lj_fluid(n_side, dtype=np.float64, rebuild_every=5)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-9 and np.abs(res["v"] - o.vel).max() < 1e-8
    assert int(res["fused"]) == 0 and int(res["prunes"]) >= 1


@pytest.mark.parametrize("fused", [0, 1])
def test_extra_checks_between_cadence_steps_are_honoured(fused, tmp_path, monkeypatch):
    """mhip_plan_decide may keep the lists but vouch for them for k < rebuild_every steps only (check_in = k): the host loop then has to
    come back k steps later — the fused loop AND the two-call loop that grids with more than two bricks per axis use (round 2: the latter
    only looked at the cadence steps and would have walked an inner list past the horizon it was validated for)."""
    monkeypatch.setenv("MOLLYHIP_HALO_FUSED", str(fused)); monkeypatch.setenv("MOLLYHIP_HOST_PRUNE", "0"); monkeypatch.setenv("MOLLYHIP_TEST_EXTRA_CHECK", "3")
    world, n_side, n_steps = 2, 10, 17
    mp.spawn(_worker, args=(world, _free_port(), n_side, n_steps, str(tmp_path), 0.3, 0.2), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    assert int(res["fused"]) == fused
    assert list(res["decide_steps"]) == [5, 8, 10, 13, 15]           # cadence 5, and three steps after each decision that asked for it
    case = S.lj_fluid(n_side, dtype=np.float64, rebuild_every=5)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-9


@pytest.mark.parametrize("gm,skin,expect", [(0.3, 0.2, "one plan"), (0.3, 0.012, "prunes"), (0.012, 0.012, "replans")])
def test_long_lived_ghost_plan_matches_single_domain_oracle(gm, skin, expect, tmp_path):
    """ghost shell r_list + margin: prunes are scheduled collectively from the displacement since the last prune (skin), and a
    prune is only allowed while nobody moved margin/2 since the plan — else the plan is redone.  (The skin passed to the host logic
    is artificial here — the stand-in engine searches at every step — it only drives the schedule under test.)"""
    world, n_side, n_steps = 2, 10, 30     # bricks 1.8 nm >= 1.2 + 0.3; rebuild cadence 5 → 6 checks
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_side, n_steps, str(tmp_path), gm, skin), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    case = S.lj_fluid(n_side, dtype=np.float64, rebuild_every=5)
    o = case.oracle(np.float64)
    o.vv_run(n_steps, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-9 and np.abs(res["v"] - o.vel).max() < 1e-8
    assert int(res["checks"]) == n_steps // 5 and int(res["fused"]) == 1
    if expect == "one plan":
        assert int(res["plans"]) == 1 and int(res["prunes"]) == 0     # 0.033 nm of drift in 30 steps: neither skin nor margin used up
    elif expect == "prunes":
        assert int(res["plans"]) == 1 and int(res["prunes"]) >= 2     # small skin, wide margin: prunes only
    else:
        assert int(res["plans"]) > 1                                   # small margin: a due prune finds the plan stale → re-plan


def _fallback_worker(rank, world, port, mode, out_dir):
    """an engine that HAS the in-engine exchange entry points but where one rank cannot use them: every rank must end up on the host loop"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import molly_loader
    molly_loader.load()
    from molly_jl_amd import domain
    from tests.oracle_domain_engine import OracleDomainEngine

    class Flaky(OracleDomainEngine):
        calls = []

        def halo_region(self, rows, w, r):
            self.calls.append("region")
            if mode == 0 and r == 1:
                raise RuntimeError("no IPC here")
            return bytes(64)

        def halo_open_peer(self, r, handle):
            self.calls.append("open")
            if mode == 1 and self.my_rank == 0:
                raise RuntimeError("cannot map")

        def halo_selftest(self):
            self.calls.append("selftest")
            return not (mode == 2 and self.my_rank == 1)

        def set_halo_routes(self, *a):
            raise AssertionError("routes must not be set once a rank has fallen back")

        def domain_run(self, *a):
            raise AssertionError("the engine loop must not run once a rank has fallen back")

    case = S.lj_fluid(10, dtype=np.float64, rebuild_every=5)
    grid = domain.choose_grid(world, case.box)
    bg = domain.BrickGrid(case.box, grid, rank, case.r_list + 0.3)
    box, origin, periodic = bg.engine_box(pad=0.3)
    eng = Flaky(case.inter_dict(np.float64), case.box, periodic, case.r_list, ghost_margin=0.3, skin=0.2, every=case.rebuild_every)
    eng.my_rank = rank
    run = domain.DomainRun(bg, eng, torch.float64, torch.device("cpu"), case.rebuild_every, ghost_margin=0.3, skin=0.2)
    assert run.engine_loop                                     # the entry points are there: it starts out wanting the engine loop
    run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
    fell = torch.tensor([0 if run.engine_loop else 1]); dist.all_reduce(fell)
    assert int(fell) == world, (mode, rank, run.engine_loop)   # EVERY rank fell back, also the ones whose own calls succeeded
    run.run(0, 12, 0.002, remove_cm_every=1)
    xs, vs = run.gather_global(case.n)
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), x=xs, v=vs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_engine_loop_falls_back_collectively(mode, tmp_path):
    """mhip_halo_region / mhip_halo_open_peer / mhip_halo_selftest failing on ONE rank (no IPC for fine-grained memory, a peer that cannot
    be mapped, a store that never becomes visible) keeps EVERY rank on the host loop with torch.distributed collectives — decided by
    all_gathers at setup, so no rank waits in a kernel for a peer that went the other way; the run is the same run"""
    world = 2
    mp.spawn(_fallback_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    res = np.load(os.path.join(tmp_path, "result.npz"))
    case = S.lj_fluid(10, dtype=np.float64, rebuild_every=5)
    o = case.oracle(np.float64)
    o.vv_run(12, 0.002, remove_cm_every=1)
    d = res["x"] - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-9 and np.abs(res["v"] - o.vel).max() < 1e-8


def test_brick_grid_geometry():
    import molly_loader
    molly_loader.load()
    from molly_jl_amd import domain
    assert domain.choose_grid(1, [36.0] * 3) == (1, 1, 1) and domain.choose_grid(2, [36.0] * 3) == (2, 1, 1)
    assert domain.choose_grid(4, [36.0] * 3) == (2, 2, 1) and domain.choose_grid(8, [36.0] * 3) == (2, 2, 2)
    g = domain.BrickGrid([36.0] * 3, (2, 2, 2), 5, 1.2)      # rank 5 = brick (1, 0, 1)
    assert g.coord == (1, 0, 1) and len(g.dirs) == 26
    peers = {p for p, _, _ in g.dirs}
    assert peers == set(range(8)) - {5}                       # 7 distinct peers, each reached through several faces/images
    # a +x neighbour of the last brick wraps: ghosts are shifted by -L
    (p, dv, sh) = [t for t in g.dirs if t[1] == (1, 0, 0)][0]
    assert p == g.rank_of((0, 0, 1)) and sh == (-36.0, 0.0, 0.0)
    x = torch.tensor([[0.1, 0.1, 0.1], [35.9, 17.9, 18.1], [18.0, 18.0, 18.0]], dtype=torch.float64)
    assert g.owner_of(x).tolist() == [0, 5, 7]
    with pytest.raises(ValueError):
        domain.BrickGrid([2.0] * 3, (2, 1, 1), 0, 1.2)        # brick narrower than the ghost reach
    box, origin, periodic = domain.BrickGrid([36.0] * 3, (2, 1, 1), 1, 1.2).engine_box(pad=0.3)
    assert periodic == [0, 1, 1] and box[0] == pytest.approx(18 + 3.0) and origin[0] == pytest.approx(18 - 1.5)

"""Langevin integrator, Andersen thermostat and random velocities on the MI355X (csrc/stochastic.hip) against the oracle's restatement
(oracle/stochastic.h) with the same Philox keys and counters, plus the known-answer vectors of the generator on the device."""
import ctypes as C

import numpy as np
import pytest

from tests import systems as S
from tests.test_oracle_stochastic import KB, PHILOX_KAT, ideal_gas

pytestmark = pytest.mark.gpu


def draws(seed, n):
    """the uint64 words simulate(..., rng=seed) draws, in its order"""
    rng = np.random.default_rng(seed)
    return [int(rng.integers(0, 2 ** 64, dtype=np.uint64)) for _ in range(n)]


@pytest.mark.parametrize("ctr,key,out", PHILOX_KAT)
def test_device_philox_known_answers(pkg, ctr, key, out):
    c = np.asarray(ctr, np.uint32); k = np.asarray(key, np.uint32); o = np.zeros(4, np.uint32)
    rc = pkg.lib().mhip_philox4x32_10(c.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    assert rc == 0 and [int(x) for x in o] == out


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-13), (np.float32, 2e-6)])
def test_random_velocities_and_andersen_match_the_oracle(pkg, dtype, tol):
    """same counters → same normals (device log / sincos against libm: a few ulp), and EXACTLY the same atoms picked by the thermostat"""
    n = 20000
    mass = np.where(np.arange(n) % 3 == 0, 1.008, 15.999); mass[11] = 0.0
    case = ideal_gas(n, dtype, mass=mass)
    ctr1, key = draws(7, 2)
    o = case.oracle(dtype); o.random_velocities(KB * 300.0, key=key, ctr1=ctr1)
    s = case.system(pkg, dtype)
    pkg.random_velocities(s, 300.0, rng=7)
    scale = np.sqrt(KB * 300.0 / 1.008)
    assert np.abs(s.velocities.astype(np.float64) - o.vel).max() < 6 * tol * scale
    assert np.all(s.velocities[11] == 0)
    v0 = s.velocities.copy()
    sim = pkg.VelocityVerlet(dt=0.002)
    assert pkg.apply_coupling(s, pkg.AndersenThermostat(250.0, 0.01), sim, rng=8) is False
    ctr1, key = draws(8, 2)
    v0_ref = o.vel.copy()
    o.andersen(KB * 250.0, 0.2, key=key, ctr1=ctr1)
    hit, hit_ref = np.any(s.velocities != v0, axis=1), np.any(o.vel != v0_ref, axis=1)
    assert np.array_equal(hit, hit_ref) and 0.17 < hit.mean() < 0.23
    assert np.abs(s.velocities.astype(np.float64) - o.vel).max() < 6 * tol * scale
    assert np.array_equal(s.velocities[~hit], v0[~hit])


def test_langevin_fp64_matches_oracle(pkg):
    """simulate!(sys, ::Langevin, n) (simulators.jl:1099-1220) on the charged fluid with exceptions, CM removal every step: 30 steps
    against the oracle with the same key / counter"""
    case = S.charged_fluid(10, dict(kind="rf", rc=1.0), dtype=np.float64, with_exceptions=True, stable=True)
    key, ctr1 = draws(21, 2)
    o = case.oracle(np.float64)
    o.langevin_run(30, 0.0005, KB * 300.0, 5.0, key=key, ctr1=ctr1, remove_cm_every=1, nthreads=4)
    s = case.system(pkg, np.float64)
    pkg.simulate(s, pkg.Langevin(dt=0.0005, temperature=300.0, friction=5.0), 30, rng=21)
    assert np.abs(s.coords - o.coords).max() < 1e-9
    assert np.abs(s.velocities - o.vel).max() < 1e-7
    assert np.abs((s.velocities * case.mass[:, None]).sum(axis=0)).max() < 1e-9       # remove_CM_motion! after the last step


def test_langevin_fp32_6mrr_pme_tracks_fp32_oracle(pkg):
    """the complete 6mrr step (pair list + bonded + PME) under the Langevin integrator over 10 steps.  The fp32 noise is its own stream
    (one Philox block per atom, single-precision Box-Muller), so the partner is the oracle's fp32 instance."""
    from tests import golden6mrr as G
    case = G.case("ewald", np.float32, bonded=True, pme=True)
    key, ctr1 = draws(5, 2)
    o = case.oracle(np.float32)
    o.langevin_run(10, 0.0005, KB * 300.0, 1.0, key=key, ctr1=ctr1, remove_cm_every=1, nthreads=8, specific=True, general=True)
    s = case.system(pkg, np.float32)
    pkg.simulate(s, pkg.Langevin(dt=0.0005, temperature=300.0, friction=1.0), 10, rng=5)
    d = s.coords.astype(np.float64) - o.coords
    d -= np.round(d / case.box) * case.box
    print("6mrr langevin fp32: max |dx|", np.abs(d).max(), "max |dv|", np.abs(s.velocities - o.vel).max())
    assert np.abs(d).max() < 2e-5 and np.abs(s.velocities - o.vel).max() < 5e-3
    assert s.stats()["n_fused_steps"] >= 9          # the steps integrated inside their last force launch (k_gather_collect_vv<…, LANG>); the first follows remove_CM_motion!'s v_cm


def test_langevin_fp64_6mrr_pme_matches_oracle(pkg):
    """the same configuration in fp64, where the device's noise IS the oracle's (two Philox blocks per atom, double-precision Box-Muller): 8 steps of the complete step with
    the update inside the last force launch against the oracle, at the fp64 trajectory bars"""
    from tests import golden6mrr as G
    case = G.case("ewald", np.float64, bonded=True, pme=True)
    key, ctr1 = draws(6, 2)
    o = case.oracle(np.float64)
    o.langevin_run(8, 0.0005, KB * 300.0, 1.0, key=key, ctr1=ctr1, remove_cm_every=1, nthreads=8, specific=True, general=True)
    s = case.system(pkg, np.float64)
    pkg.simulate(s, pkg.Langevin(dt=0.0005, temperature=300.0, friction=1.0), 8, rng=6)
    d = s.coords - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-9 and np.abs(s.velocities - o.vel).max() < 1e-6
    assert s.stats()["n_fused_steps"] >= 7


@pytest.mark.parametrize("remove_cm,andersen", [(1, False), (0, False), (1, True)])
def test_langevin_inside_the_last_force_launch_is_the_same_run(pkg, monkeypatch, remove_cm, andersen):
    """mhip_langevin_run on the complete PME configuration: every step's last force launch (interpolation + bonded sums) runs the Langevin-middle update too
    (step_fused.h; the same langevin_atom the stand-alone kernel calls, the same Philox words) instead of k_gather_collect + k_langevin.  40 steps across rebuilds and a
    chunk boundary against the two-launch form (MOLLYHIP_FUSE_GATHER_VV=0): fp32 round-off apart (the charge mesh is flushed with float atomics)."""
    from tests import golden6mrr as G

    def run(fuse):
        monkeypatch.setenv("MOLLYHIP_FUSE_GATHER_VV", fuse)
        case = G.case("ewald", np.float32, bonded=True, pme=True)
        s = case.system(pkg, np.float32)
        sim = pkg.Langevin(dt=0.0005, temperature=300.0, friction=1.0, remove_CM_motion=remove_cm,
                           coupling=pkg.AndersenThermostat(300.0, 0.05) if andersen else None)
        pkg.simulate(s, sim, 25, rng=9)
        pkg.simulate(s, sim, 15, init_step=25, rng=10)
        return np.array(s.coords, dtype=np.float64), np.array(s.velocities, dtype=np.float64), s.stats()
    x1, v1, st1 = run("1")
    x0, v0, st0 = run("0")
    assert st1["n_fused_steps"] >= 36 and st0["n_fused_steps"] == 0, (st1["n_fused_steps"], st0["n_fused_steps"])
    box = G.data()["box"]
    d = x1 - x0; d -= np.round(d / box) * box
    # (measured: 1.9e-6 nm = four fp32 ulps of a 6 nm coordinate, 1.0e-3 nm/ps — tools/micro/langevin_dev_check.py; the noise enters through the velocities, so the bar on the
    # coordinates sits a little above the velocity-Verlet twin's 4e-6, tests/test_gpu_6mrr.py)
    assert np.abs(d).max() < 1e-5 and np.abs(v1 - v0).max() < 4e-3, (np.abs(d).max(), np.abs(v1 - v0).max())


def test_langevin_list_checks_measured_by_the_update_launch(pkg, monkeypatch):
    """inside mhip_langevin_run of the complete PME configuration the validity checks of the pair lists are measured by the launch that makes the coordinates and read two
    steps later (no drained stream), as inside mhip_vv_run; the two-launch form (MOLLYHIP_FUSE_GATHER_VV=0) keeps the drained check at the check step itself.  fp64, 90 steps in
    two chunks across nine cadence steps at 350 K: a pair missing from a list for a single pass would show as a force jump — the runs agree to round-off"""
    from tests import golden6mrr as G

    def run(fuse):
        monkeypatch.setenv("MOLLYHIP_FUSE_GATHER_VV", fuse)
        case = G.case("ewald", np.float64, bonded=True, pme=True)
        s = case.system(pkg, np.float64)
        sim = pkg.Langevin(dt=0.001, temperature=350.0, friction=2.0, remove_CM_motion=1)
        pkg.simulate(s, sim, 55, rng=12)
        pkg.simulate(s, sim, 35, init_step=55, rng=13)
        return np.array(s.coords), np.array(s.velocities), s.stats()
    x1, v1, st1 = run("1")
    x0, v0, st0 = run("0")
    assert st1["n_fused_steps"] >= 85 and st0["n_fused_steps"] == 0
    assert st1["n_filter_passes"] >= 2 and st0["n_filter_passes"] >= 2, (st1["n_filter_passes"], st0["n_filter_passes"])
    box = G.data()["box"]
    d = x1 - x0; d -= np.round(d / box) * box
    assert np.abs(d).max() < 1e-8 and np.abs(v1 - v0).max() < 1e-6, (np.abs(d).max(), np.abs(v1 - v0).max())


@pytest.mark.parametrize("remove_cm", [1, 0])
def test_langevin_in_the_pair_pass_epilogue_of_the_lj_fluid(pkg, monkeypatch, remove_cm):
    """mhip_langevin_run on the fp32 one-type fluids: the plain pair passes run the Langevin-middle update in their own epilogue (k_forces<…, STEP, ·, LANG>: the force
    still in registers, langevin_atom of philox.h — the function the stand-alone kernel calls —, v_cm of the step before published by the head workgroup, list checks
    measured on the way) — no force array, no k_langevin launch.  64 000 atoms, 80 steps in two chunks across searches, prunes and checks, against the two-launch form
    (MOLLYHIP_FUSE_STEP=0: its checks and therefore its prunes fall on other steps, so fp32 round-off apart rather than bit for bit) and, over the first 20 steps,
    against the fp32 oracle with the same Philox words."""
    case = S.lj_fluid(40, seed=2, dtype=np.float32)
    sim = pkg.Langevin(dt=0.002, temperature=85.0, friction=1.0, remove_CM_motion=remove_cm)
    out = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("MOLLYHIP_FUSE_STEP", fuse)      # (read when the engine is created)
        s = case.system(pkg, np.float32)
        pkg.simulate(s, sim, 20, rng=31)
        first = (s.coords.copy(), s.velocities.copy())
        pkg.simulate(s, sim, 60, init_step=20, rng=32)
        st = s.stats()
        assert (st["n_fused_steps"] > 60) == (fuse == "1"), st["n_fused_steps"]
        out.append((s.coords.astype(np.float64), s.velocities.astype(np.float64), first))
        s.close()
    d = out[0][0] - out[1][0]; d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 2e-5 and np.abs(out[0][1] - out[1][1]).max() < 5e-3, (np.abs(d).max(), np.abs(out[0][1] - out[1][1]).max())
    key, ctr1 = draws(31, 2)
    o = case.oracle(np.float32)
    o.langevin_run(20, 0.002, KB * 85.0, 1.0, key=key, ctr1=ctr1, remove_cm_every=remove_cm, nthreads=8)
    d = out[0][2][0].astype(np.float64) - o.coords; d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 2e-5 and np.abs(out[0][2][1] - o.vel).max() < 5e-3, (np.abs(d).max(), np.abs(out[0][2][1] - o.vel).max())


def test_langevin_is_reproducible_and_chunks_continue(pkg):
    """counter-based noise: two runs with the same rng are bit-identical in fp64 coordinates up to summation order (none here: no
    atomics on the pair path), and different seeds decorrelate"""
    case = S.lj_fluid(10, dtype=np.float32)
    sim = pkg.Langevin(dt=0.002, temperature=85.0, friction=2.0)
    a = case.system(pkg, np.float32); pkg.simulate(a, sim, 25, rng=3)
    b = case.system(pkg, np.float32); pkg.simulate(b, sim, 25, rng=3)
    assert np.array_equal(a.coords, b.coords) and np.array_equal(a.velocities, b.velocities)
    c = case.system(pkg, np.float32); pkg.simulate(c, sim, 25, rng=4)
    assert np.abs(a.velocities - c.velocities).max() > 1e-3


def test_langevin_thermalises_the_lj_fluid(pkg):
    """test/simulation.jl: Langevin runs are judged by their temperature; 85 K argon driven to 140 K within ~10 relaxation times"""
    case = S.lj_fluid(16, dtype=np.float32)
    s = case.system(pkg, np.float32)
    pkg.simulate(s, pkg.Langevin(dt=0.002, temperature=140.0, friction=5.0), 1000, rng=1)
    ts = []
    for k in range(10):
        pkg.simulate(s, pkg.Langevin(dt=0.002, temperature=140.0, friction=5.0), 50, init_step=1000 + 50 * k, rng=100 + k)
        ts.append(pkg.temperature(s))
    assert abs(np.mean(ts) - 140.0) < 0.04 * 140.0
    assert np.isfinite(s.coords).all()


def test_velocity_verlet_with_andersen_coupling_matches_oracle(pkg):
    """AndersenThermostat as the coupling of VelocityVerlet (simulators.jl:630): same atoms re-drawn at the same steps as the oracle"""
    case = S.lj_fluid(8, dtype=np.float64)
    (seed,) = draws(13, 1)
    o = case.oracle(np.float64)
    o.set_andersen(KB * 300.0, 0.002 / 0.05, seed=seed)
    o.vv_run(20, 0.002, remove_cm_every=1)
    s = case.system(pkg, np.float64)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002, coupling=pkg.AndersenThermostat(300.0, 0.05)), 20, rng=13)
    assert np.abs(s.coords - o.coords).max() < 1e-9 and np.abs(s.velocities - o.vel).max() < 1e-8
    plain = case.system(pkg, np.float64)
    pkg.simulate(plain, pkg.VelocityVerlet(dt=0.002), 20)
    assert np.abs(plain.velocities - s.velocities).max() > 0.05     # the thermostat really acted; and it is off again afterwards:
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), 1, init_step=20)
    o.set_andersen(0, 0, 0); o.vv_run(1, 0.002, first_step=20, remove_cm_every=1)
    assert np.abs(s.velocities - o.vel).max() < 1e-8


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reference_bars_for_device_noise(pkg, dtype):
    """test/gpu_consistency.jl:587-679 at its own sizes and bars: 50 000 atoms of mass 10 at 298 K, three massless (virtual-site) atoms whose
    velocities must be actively zeroed — device against host within 16 eps(FT)·sqrt(kT/m); and one O-step with vel_scale 0.8,
    noise_scale 0.3, key 0x1234567890abcdef, ctr1 0xfedcba0987654321 on N(0,1) velocities within 16 eps(FT)"""
    n = 50000
    eps = np.finfo(dtype).eps
    mass = np.full(n, 10.0); mass[[1, 3, 6]] = 0.0
    case = ideal_gas(n, dtype, mass=mass)
    case.velocities = np.ones((n, 3))                           # the buffers start non-zero (:624-627)
    ctr1, key = draws(10, 2)
    o = case.oracle(dtype); o.random_velocities(KB * 298.0, key=key, ctr1=ctr1)
    s = case.system(pkg, dtype)
    pkg.random_velocities(s, 298.0, rng=10)
    vel_scale = np.sqrt(KB * 298.0 / 10.0)
    assert np.linalg.norm(s.velocities.astype(np.float64) - o.vel, axis=1).max() < 16 * eps * vel_scale
    assert np.all(s.velocities[[1, 3, 6]] == 0) and np.all(o.vel[[1, 3, 6]] == 0)

    gas = ideal_gas(n, dtype, mass=np.ones(n))
    gas.velocities = np.random.default_rng(15).normal(size=(n, 3)).astype(dtype).astype(np.float64)
    dt = 0.001
    friction, kT = -np.log(0.8) / dt, 0.25                      # exp(−γ dt) = 0.8 ; sqrt(1 − 0.8²)·sqrt(kT/m) = 0.3
    o = gas.oracle(dtype)
    o.langevin_run(1, dt, kT, friction, key=0x1234567890abcdef, ctr1=0xfedcba0987654321, remove_cm_every=0)
    s = gas.system(pkg, dtype)
    s.push_state(velocities=True)
    s._check(pkg.lib().mhip_langevin_run(s._ctx, 0, 1, dt, kT, friction, 0, 0x1234567890abcdef, 0xfedcba0987654321))
    s.pull_state()
    assert not np.array_equal(s.velocities, gas.velocities.astype(dtype))
    assert np.linalg.norm(s.velocities.astype(np.float64) - o.vel, axis=1).max() < 16 * eps

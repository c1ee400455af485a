"""The boundary as a variable: `sys.boundary = …` on a live system (scale_coords!, spatial.jl:1184-1210; the barostats of coupling.jl:861-1033).  The reference's
force / energy entry points read sys.boundary at every call (ext/MollyCUDAExt.jl:845, 936), so the engine's context must follow the box (mhip_set_box): the same
numbers as a context created on the new box, the oracle's numbers on the new box, and the old numbers again after a rejected move."""
import copy
import math

import numpy as np
import pytest

from tests import systems as S
from tests.test_gpu_triclinic import sheared_fluid

pytestmark = pytest.mark.gpu


def on_box(case, dtype, coords, box, basis=None):
    """the same atoms and interactions as a case of their own on another box (the PME mesh is the constructor's, ewald.jl:285-309: it does not follow the box)"""
    c = copy.copy(case)
    c.coords = np.asarray(coords, dtype=np.float64).reshape(-1, 3).copy()
    c.box = np.asarray(box, dtype=np.float64).copy()
    if case.pme is not None:
        c.pme = dict(case.pme, mesh=case.pme_params(dtype)["mesh"])
    if basis is not None:
        c.triclinic = dict(case.triclinic, basis=np.asarray(basis, dtype=np.float64))
    return c


def make(kind):
    if kind == "lj_fp64":
        return S.lj_fluid(12, dtype=np.float64), np.float64
    if kind == "lj_fp32_packed":           # 64 000 atoms: the packed one-type kernels, dual list, 128-atom blocks
        return S.lj_fluid(40, dtype=np.float32), np.float32
    if kind == "pme_fp64":
        return S.charged_fluid(10, dict(kind="ewald", rc=1.0, tol=5e-4), dtype=np.float64, pme=dict(order=5, error_tol=5e-4)), np.float64
    if kind == "pme_fp32":
        return S.charged_fluid(10, dict(kind="ewald", rc=1.0, tol=5e-4), dtype=np.float32, pme=dict(order=5, error_tol=5e-4)), np.float32
    if kind == "triclinic_fp64":
        return sheared_fluid(np.float64)[1], np.float64
    raise KeyError(kind)


def everything(pkg, s):
    full = dict(specific=bool(s.specific_inter_lists), general=bool(s.general_inters))
    nl = pkg.find_neighbors(s)
    return pkg.forces(s, **full).astype(np.float64), pkg.potential_energy(s, **full), pkg.virial(s, **full), S.sorted_pairs(nl.i, nl.j, nl.special)


@pytest.mark.parametrize("kind", ["lj_fp64", "lj_fp32_packed", "pme_fp64", "pme_fp32", "triclinic_fp64"])
def test_a_live_context_follows_the_boundary(pkg, kind):
    case, dtype = make(kind)
    s = case.system(pkg, dtype)
    f0, e0, w0, nl0 = everything(pkg, s)                     # lists built, forces cached on the old box
    x_old, b_old = s.coords.copy(), s.boundary
    mu = np.diag([1.02, 0.97, 1.015]) if kind != "triclinic_fp64" else np.diag([1.02, 1.02, 1.02])
    pkg.scale_coords(s, mu)
    assert s._box_dirty and s.boundary is not b_old
    f1, e1, w1, nl1 = everything(pkg, s)
    assert not s._box_dirty
    # … a context created on the new box
    basis = s.boundary.basis_vectors if kind == "triclinic_fp64" else None
    case2 = on_box(case, dtype, s.coords, s.boundary.side_lengths, basis)
    s2 = case2.system(pkg, dtype)
    f2, e2, w2, nl2 = everything(pkg, s2)
    assert all(np.array_equal(u, v) for u, v in zip(nl1, nl2)) and len(nl1[0]) != len(nl0[0])
    fmax = np.abs(f2).max()
    tight = 1e-11 if dtype == np.float64 else 2e-5
    assert np.abs(f1 - f2).max() <= tight * fmax and abs(e1 - e2) <= tight * abs(e2) and np.abs(w1 - w2).max() <= tight * np.abs(w2).max()
    # … the oracle on the new box, at the bars of the parity tests
    o = case2.oracle(np.float64)
    onl = o.neighbors("brute" if kind == "triclinic_fp64" else "cell", nthreads=8)
    full = dict(specific=bool(s.specific_inter_lists), general=bool(s.general_inters))
    f_ref, e_ref = o.forces(onl, nthreads=4, **full).astype(np.float64), o.potential_energy(onl, **full)
    if dtype == np.float64:
        assert np.abs(f1 - f_ref).max() < 1e-8 * np.abs(f_ref).max() and e1 == pytest.approx(e_ref, rel=1e-9)
        o64 = case2.oracle(np.float64).neighbors("brute" if kind == "triclinic_fp64" else "cell", nthreads=8)
        assert all(np.array_equal(u, v) for u, v in zip(nl1, S.sorted_pairs(*o64)))
    else:
        scale, jump = o.force_scale(onl)
        extra = 2e-4 * np.linalg.norm(f_ref, axis=1).max() if kind == "pme_fp32" else 0.0      # reciprocal space in single precision (tests/test_gpu_pme.py)
        assert np.all(np.linalg.norm(f1 - f_ref, axis=1) <= 4e-5 * scale + 1.01 * jump + 1e-4 + extra)
        assert e1 == pytest.approx(e_ref, rel=3e-5)
    # … and the old numbers after the move is taken back (coupling.jl:929-930)
    s.coords[:] = x_old
    s.boundary = b_old
    f3, e3, w3, nl3 = everything(pkg, s)
    assert all(np.array_equal(u, v) for u, v in zip(nl3, nl0))
    assert np.abs(f3 - f0).max() <= tight * np.abs(f0).max() and abs(e3 - e0) <= tight * abs(e0)
    assert s.stats()["n_box_changes"] == 2


def test_a_run_continues_on_the_new_box(pkg):
    """velocity Verlet across a box change against the oracle doing the same: 15 steps, the box and coordinates scaled by 1.5 %, 15 steps (fp64; the velocities stay
    the engine's own across mhip_set_box)"""
    case, dtype = make("lj_fp64")
    s = case.system(pkg, dtype)
    o = case.oracle(np.float64)
    sim = pkg.VelocityVerlet(dt=0.002, remove_CM_motion=1)
    pkg.simulate(s, sim, 15)
    o.vv_run(15, 0.002)
    assert np.abs(s.coords - o.coords).max() < 1e-9
    pkg.scale_coords(s, np.diag([1.015] * 3))
    o.coords *= 1.015
    o.set_boundary(o.box * 1.015)
    assert np.allclose(s.boundary.side_lengths, o.box, rtol=0, atol=1e-14)
    pkg.simulate(s, sim, 15, init_step=15)
    o.vv_run(15, 0.002, first_step=15)
    assert np.abs(s.coords - o.coords).max() < 1e-9 and np.abs(s.velocities - o.vel).max() < 1e-8
    assert s.stats()["n_box_changes"] == 1


def mc_isotropic_reference(energy, coords, box, rng, volume_scale, pressure_bar, temperature, n_atoms, kB, bar):
    """one attempt of apply_coupling_mc!(…, Val(:isotropic), …) restated from coupling.jl:886-932 (fp64, no topology: every atom a molecule):
    → (accepted, coords, box)"""
    kT = kB * temperature
    E = energy(coords, box)
    V = float(np.prod(box))
    dV = volume_scale * (2 * rng.random() - 1)
    v_scale = (V + dV) / V
    l_scale = np.cbrt(v_scale)
    E_trial = energy(coords * l_scale, box * l_scale)
    dW = (E_trial - E) + 3 * pressure_bar * bar * dV / 3 - n_atoms * kT * math.log(v_scale)
    if dW <= 0 or rng.random() < math.exp(-dW / kT):
        return True, coords * l_scale, box * l_scale
    return False, coords, box


def test_monte_carlo_barostat_decisions_match_the_oracle(pkg):
    """MonteCarloBarostat (coupling.jl:861-932) through the engine: twelve applications on a charged fluid with PME and exception lists, the same uniform numbers
    replayed over the oracle's potential energies — the same accept / reject sequence (both outcomes occur), the same box, the adapted volume scale"""
    case, dtype = make("pme_fp64")
    s = case.system(pkg, dtype)
    baro = pkg.MonteCarloBarostat(1.0, 300.0, s.boundary, n_steps=1, scale_factor=0.004)
    vs0 = baro.volume_scale
    o = case.oracle(np.float64)

    def energy(x, box):
        o.coords[:] = x
        o.set_boundary(box)
        return o.potential_energy(o.neighbors("cell", nthreads=8), specific=True, general=True)

    rng_a, rng_b = np.random.default_rng(77), np.random.default_rng(77)
    x, box = case.coords.copy(), case.box.copy()
    got, want = [], []
    n_att = n_acc = 0; vs = vs0
    for step in range(1, 13):
        got.append(pkg.apply_coupling(s, baro, None, rng=rng_a, step_n=step))
        acc, x, box = mc_isotropic_reference(energy, x, box, rng_b, vs, 1.0, 300.0, case.n, pkg.BOLTZMANN, pkg.BAR)
        want.append(acc)
        n_att += 1; n_acc += int(acc)
        if n_att >= 10:                                          # coupling.jl:871-881
            if n_acc < 0.25 * n_att:
                vs /= 1.1
            elif n_acc > 0.75 * n_att:
                vs = min(vs * 1.1, float(np.prod(box)) * 0.3)
            n_att = n_acc = 0
        assert np.allclose(s.boundary.side_lengths, box, rtol=1e-13, atol=0), step
        assert np.abs(s.coords - x).max() < 1e-12, step
    assert got == want and True in got and False in got, (got, want)
    assert baro.volume_scale == pytest.approx(vs, rel=1e-14)
    assert s.stats()["n_box_changes"] >= 12


def test_langevin_with_a_monte_carlo_barostat(pkg):
    """the README's GPU example in small (README.md:126-133: Langevin + MonteCarloBarostat): 40 steps with the barostat every 10, against the oracle's Langevin chunks with
    the replayed barostat in between — same random words, same trajectory (fp64)"""
    case, dtype = make("lj_fp64")
    s = case.system(pkg, dtype)
    T0, dt, fric = 85.0, 0.002, 1.0
    baro = pkg.MonteCarloBarostat(1.0, T0, s.boundary, n_steps=10, scale_factor=0.002)
    sim = pkg.Langevin(dt=dt, temperature=T0, friction=fric, coupling=baro, remove_CM_motion=1)
    pkg.simulate(s, sim, 40, rng=np.random.default_rng(5))
    rng = np.random.default_rng(5)
    key, ctr1 = int(rng.integers(0, 2 ** 64, dtype=np.uint64)), int(rng.integers(0, 2 ** 64, dtype=np.uint64))
    o = case.oracle(np.float64)

    def energy(x, box):
        keep = o.coords.copy()
        o.coords[:] = x
        o.set_boundary(box)
        e = o.potential_energy(o.neighbors("cell", nthreads=8))
        o.coords[:] = keep
        return e

    accepted = []
    for first in range(0, 40, 10):
        o.langevin_run(10, dt, pkg.BOLTZMANN * T0, fric, key, (ctr1 + first) % 2 ** 64, first_step=first)
        acc, x, box = mc_isotropic_reference(energy, o.coords.copy(), o.box.copy(), rng, baro_scale(baro, accepted), 1.0, T0, case.n, pkg.BOLTZMANN, pkg.BAR)
        accepted.append(acc)
        o.coords[:] = x
        o.set_boundary(box)
    assert np.allclose(s.boundary.side_lengths, o.box, rtol=1e-12, atol=0)
    assert np.abs(s.coords - o.coords).max() < 1e-7 and np.abs(s.velocities - o.vel).max() < 1e-6
    assert s.stats()["n_box_changes"] >= 4


def baro_scale(baro, accepted):
    """(fewer than ten attempts: the volume scale is still the constructor's)"""
    assert len(accepted) < 10
    return baro.volume_scale


def test_set_box_refusals(pkg):
    import ctypes as C
    case, dtype = make("lj_fp64")
    s = case.system(pkg, dtype)
    pkg.forces(s)
    L = pkg.lib()
    box = (C.c_double * 3)(4.0, 4.0, -1.0)
    assert L.mhip_set_box(s.engine(), box, None) == -1 and b"positive" in L.mhip_last_error(s.engine())                  # MHIP_ERR_INVALID
    bv = (C.c_double * 9)(4, 0, 0, 0, 4, 0, 0, 0, 4)
    box = (C.c_double * 3)(4.0, 4.0, 4.0)
    assert L.mhip_set_box(s.engine(), box, bv) == -1                                                                     # a basis for a CubicBoundary
    assert L.mhip_set_box(s.engine(), None, None) == -1
    with pytest.raises(pkg.MollyHipError):
        s.boundary = pkg.TriclinicBoundary((4, 0, 0), (0, 4, 0), (0, 0, 4))                                             # a live system keeps its kind of boundary
    # the context is still good, on its old box
    f = pkg.forces(s)
    assert np.isfinite(f).all()
    # forces without handing the coordinates over again: refused
    box = (C.c_double * 3)(*[float(v) * 1.01 for v in s.boundary.side_lengths])
    assert L.mhip_set_box(s.engine(), box, None) == 0
    out = np.zeros((case.n, 3), dtype)
    assert L.mhip_forces(s.engine(), 0, 0, s._ptr(out), None, 0) == -3                                                   # MHIP_ERR_STATE


def test_6mrr_pme_step_on_a_scaled_box(pkg):
    """the timed 6mrr configuration (fp32, PME, bonded terms, the group-split pair launch and the step whose last force launch integrates) across a box change: forces,
    energy and twenty steps on the scaled box equal those of a context created there; the total force against the fp64 oracle at the bar of tests/test_gpu_pme.py"""
    from tests import golden6mrr as G
    case = G.case("ewald", np.float32, pme=True)
    s = case.system(pkg, np.float32)
    sim = pkg.VelocityVerlet(dt=0.0005, remove_CM_motion=1)
    pkg.simulate(s, sim, 10)                                  # lists, group-split launch, fused steps: all live on the old box
    assert s.stats()["n_group_split_passes"] > 0
    pkg.scale_coords(s, np.diag([1.004, 1.004, 1.004]))
    full = dict(specific=True, general=True)
    f1, e1 = pkg.forces(s, **full).astype(np.float64), pkg.potential_energy(s, **full)
    case2 = on_box(case, np.float32, s.coords, s.boundary.side_lengths)
    s2 = case2.system(pkg, np.float32, velocities=s.velocities)
    f2, e2 = pkg.forces(s2, **full).astype(np.float64), pkg.potential_energy(s2, **full)
    fmax = np.linalg.norm(f2, axis=1).max()
    assert np.linalg.norm(f1 - f2, axis=1).max() <= 2e-5 * fmax and abs(e1 - e2) <= 2e-6 * abs(e2)
    o = case2.oracle(np.float64)
    f_ref = o.forces(o.neighbors("cell", nthreads=8), nthreads=8, specific=True, general=True).astype(np.float64)
    err = np.linalg.norm(f1 - f_ref, axis=1)
    assert np.sqrt((err ** 2).mean()) < 3e-5 * np.sqrt((f_ref ** 2).sum(axis=1).mean()) and err.max() < 1e-3 * fmax
    n0 = s.stats()["n_fused_steps"]
    pkg.simulate(s, sim, 20, init_step=10)
    pkg.simulate(s2, sim, 20, init_step=10)
    assert s.stats()["n_fused_steps"] > n0
    assert np.abs(s.coords - s2.coords).max() < 5e-5 and np.abs(s.velocities - s2.velocities).max() < 8e-3      # (fp32 trajectory bars of tests/test_gpu_pme.py)


def test_steepest_descent_minimizer_follows_the_oracle(pkg):
    """SteepestDescentMinimizer (simulators.jl:183-271; the first call of the README's GPU example) on a charged fluid with PME and exception lists, fp64: the same
    accept / reject sequence and coordinates as the loop restated over the oracle's forces and energies, and the energy falls"""
    case, dtype = make("pme_fp64")
    rng = np.random.default_rng(3)
    x0 = case.coords + rng.normal(scale=0.01, size=case.coords.shape)              # off the lattice: forces to relax
    s = case.system(pkg, dtype, coords=x0)
    e0 = pkg.potential_energy(s)
    lines = []

    class Log:
        def write(self, t):
            lines.append(t)
    pkg.simulate(s, pkg.SteepestDescentMinimizer(step_size=0.01, max_steps=12, tol=1.0, log_stream=Log()))
    got = [w for w in "".join(lines).split() if w in ("accepted", "rejected")]
    # the loop over the oracle (restated from simulators.jl:225-262)
    o = case.oracle(np.float64, coords=x0)
    full = dict(specific=True, general=True)
    pe = lambda: o.potential_energy(o.neighbors("cell", nthreads=8), **full)
    o.wrap()
    E, hn, want = pe(), 0.01, []
    for _ in range(12):
        F = o.forces(o.neighbors("cell", nthreads=8), nthreads=4, **full)
        keep = o.coords.copy()
        o.coords += hn * F / np.linalg.norm(F, axis=1).max()
        o.wrap()
        E_trial = pe()
        if E_trial < E:
            hn, E = 6 * hn / 5, E_trial; want.append("accepted")
        else:
            o.coords[:] = keep; hn /= 5; want.append("rejected")
    assert got == want and "accepted" in got, (got, want)
    assert np.abs(s.coords - o.coords).max() < 1e-9
    assert pkg.potential_energy(s) < e0 and pkg.potential_energy(s) == pytest.approx(E, rel=1e-9)


def test_readme_gpu_example_runs(tmp_path):
    """examples/readme_6mrr_npt.py — the reference README's GPU example call by call — in small: 20 minimiser steps, 90 Langevin steps with three barostat trials"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "readme_6mrr_npt.py"), "--steps", "90", "--minimize-steps", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["finite"] and d["pe_minimized_kj_mol"] < d["pe_before_kj_mol"] and d["barostat_trials"] == 3 and d["box_changes"] >= 3
    assert 150.0 < d["temperature_K"] < 450.0 and abs(d["volume_nm3"][1] / d["volume_nm3"][0] - 1.0) < 0.05


def test_velocity_verlet_with_andersen_and_barostat_couplings(pkg):
    """a tuple of couplings as the reference takes it (coupling.jl:25-35): the AndersenThermostat inside the engine's loop, the MonteCarloBarostat between its chunks —
    40 velocity-Verlet steps of the LJ fluid (fp64) with a trial every 10: the box follows every accepted move, the run stays finite and thermal"""
    case, dtype = make("lj_fp64")
    s = case.system(pkg, dtype)
    baro = pkg.MonteCarloBarostat(1.0, 85.0, s.boundary, n_steps=10, scale_factor=0.002)
    sim = pkg.VelocityVerlet(dt=0.002, coupling=(pkg.AndersenThermostat(85.0, 0.1), baro), remove_CM_motion=1)
    v0 = pkg.volume(s.boundary)
    pkg.simulate(s, sim, 40, rng=np.random.default_rng(8))
    st = s.stats()
    assert baro.n_attempted == 4 and st["n_box_changes"] >= 4
    assert np.isfinite(s.coords).all() and 40.0 < pkg.temperature(s) < 200.0
    assert abs(pkg.volume(s.boundary) / v0 - 1.0) < 0.02
    # the engine's box is the system's: the energy of a context created on the final box agrees
    s2 = on_box(case, dtype, s.coords, s.boundary.side_lengths).system(pkg, dtype)
    assert pkg.potential_energy(s) == pytest.approx(pkg.potential_energy(s2), rel=1e-11)
    with pytest.raises(pkg.MollyHipError):
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.002, coupling=(baro, baro)), 10)                     # one barostat per simulator

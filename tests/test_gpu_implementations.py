"""test/simulation.jl:1133-1255 "Different implementations": 200 harmonic diatomics (k = 10 000 kJ mol⁻¹ nm⁻², r0 = 0.2 nm) with
Lennard-Jones σ = 0.2 nm, ϵ = 0.2 kJ/mol, DistanceCutoff 1.0 nm, mass 10, velocities at 1 K, VelocityVerlet dt = 0.02 ps for 200 steps, in
CubicBoundary(6 nm) and in TriclinicBoundary((5,0,0), (2,6,0), (3,4,7)).  Every (neighbour finder × float type) combination must end
within 1e-4 nm per coordinate of the single-thread fp64 no-list CPU run and start within 5e-4 kJ/mol of its potential energy."""
import numpy as np
import pytest

from tests import systems as S

pytestmark = pytest.mark.gpu

TRI = np.array([[5.0, 0.0, 0.0], [2.0, 6.0, 0.0], [3.0, 4.0, 7.0]])


def place_diatomics(n_mol, basis, bond, min_dist, rng):
    """place_diatomics (setup.jl:63-100): molecules of two atoms `bond` apart along x, every atom at least min_dist from all others"""
    def images(d):
        f = np.linalg.solve(basis.T, d.T).T
        return (f - np.round(f)) @ basis
    pts = np.empty((0, 3))
    while len(pts) < 2 * n_mol:
        a = rng.uniform(0, 1, 3) @ basis
        b = a + np.array([bond, 0.0, 0.0])
        ok = True
        for c in (a, b):
            if len(pts) and np.sqrt((images(pts - c) ** 2).sum(axis=1).min()) <= min_dist:
                ok = False
        if ok:
            pts = np.vstack([pts, a, b])
    return pts


def make(triclinic, use_list, r_list=1.0):
    rng = np.random.default_rng(7 if triclinic else 6)
    basis = TRI if triclinic else np.diag([6.0, 6.0, 6.0])
    n = 400
    x = place_diatomics(n // 2, basis, 0.2, 0.2, rng)
    x = x.astype(np.float32).astype(np.float64)                       # every precision starts from the same numbers
    v = (np.random.default_rng(8).normal(size=(n, 3)) * np.sqrt(8.314462618e-3 * 1.0 / 10.0)).astype(np.float32).astype(np.float64)
    i = np.arange(0, n, 2)
    return basis, S.Case(x, np.diag(basis), lj=dict(cutoff=("distance", 1.0)), r_list=r_list if use_list else np.inf, rebuild_every=10, velocities=v,
                         sigma=np.full(n, 0.2), eps=np.full(n, 0.2), mass=np.full(n, 10.0),
                         bonds=dict(i=i, j=i + 1, k=np.full(len(i), 10000.0), r0=np.full(len(i), 0.2)),
                         triclinic=dict(basis=basis) if triclinic else None, name="diatomics")


@pytest.mark.parametrize("triclinic", [False, True])
def test_different_implementations_agree(pkg, triclinic):
    basis, ref_case = make(triclinic, use_list=False)
    o = ref_case.oracle(np.float64)
    e_ref = o.potential_energy(None, pairwise=True, specific=True)
    o.vv_run(200, 0.02, remove_cm_every=1, specific=True)
    for use_list, r_list, dtype in [(False, 0, np.float64), (False, 0, np.float32), (True, 1.0, np.float64), (True, 1.0, np.float32),
                                    (True, 1.5, np.float64), (True, 1.5, np.float32)]:
        _, case = make(triclinic, use_list, r_list)
        s = case.system(pkg, dtype)
        e0 = pkg.potential_energy(s)
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.02), 200)
        d = np.linalg.solve(basis.T, (s.coords.astype(np.float64) - o.coords).T).T
        d = (d - np.round(d)) @ basis
        assert np.abs(d).sum() / (3 * case.n) < 1e-4, (use_list, r_list, dtype)
        assert abs(e0 - e_ref) < 5e-4, (use_list, r_list, dtype)


def place_atoms(n, box, min_dist, rng):
    pts = np.empty((0, 3))
    while len(pts) < n:
        c = rng.uniform(0, box, 3)
        d = pts - c
        d -= np.round(d / box) * box
        if len(pts) == 0 or (d ** 2).sum(axis=1).min() > min_dist ** 2:
            pts = np.vstack([pts, c])
    return pts


LJ_VARIANTS = [(("none",), True), (("none",), False), (("distance", 1.0), True), (("shifted_potential", 1.0), True), (("shifted_force", 1.0), True),
               (("cubic_spline", 1.0, 0.6), True)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cutoff,use_list", LJ_VARIANTS)
def test_lj_on_cpu_and_gpu(pkg, cutoff, use_list, dtype):
    """test/simulation.jl:559-629: 100 atoms (σ 0.2, ϵ 0.2, mass 10) in a 2 nm box, six LennardJones variants — NoCutoff over a 1.2 nm
    neighbour list (the result then depends on list membership and on the 10-step cadence), NoCutoff without a list, and the four cutoff
    strategies — 100 VelocityVerlet steps of 2 fs.  Bars of the reference: |ΔE| at the start < 5e-4, mean |Δx| < 5e-4 nm, final |ΔE| < 5e-3 kJ/mol"""
    rng = np.random.default_rng(12)
    n, box = 100, 2.0
    x = place_atoms(n, box, 0.2, rng).astype(np.float32).astype(np.float64)
    v = (rng.normal(size=(n, 3)) * np.sqrt(8.314462618e-3 * 298.0 / 10.0) * 0.01).astype(np.float32).astype(np.float64)
    case = S.Case(x, box, lj=dict(cutoff=cutoff), r_list=1.2 if use_list else np.inf, rebuild_every=10, velocities=v, sigma=np.full(n, 0.2), eps=np.full(n, 0.2),
                  mass=np.full(n, 10.0), name="lj100")
    o = case.oracle(np.float64)
    e0_ref = o.potential_energy(o.neighbors("brute") if use_list else None)
    o.vv_run(100, 0.002, remove_cm_every=1)
    e1_ref = o.potential_energy(o.neighbors("brute") if use_list else None)
    s = case.system(pkg, dtype)
    assert abs(pkg.potential_energy(s) - e0_ref) < 5e-4
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), 100)
    d = s.coords.astype(np.float64) - o.coords
    d -= np.round(d / box) * box
    assert np.abs(d).sum() / (3 * n) < 5e-4
    assert abs(pkg.potential_energy(s) - e1_ref) < 5e-3

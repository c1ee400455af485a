"""The oracle's PME reciprocal-space restatement (oracle/pme.h ≙ src/interactions/ewald.jl:311-929) pinned to the reference's
own OpenMM fixtures for 6mrr (test/protein.jl:208-299): total forces and energy with `nonbonded_method=:pme`, exact and
approximate erfc, and the 100-step velocity-Verlet trajectory."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests import golden6mrr as G


def test_mesh_and_alpha_match_the_reference_constructor():
    case = G.case("ewald", np.float64, pme=True)
    p = case.pme_params(np.float64)
    assert p["mesh"] == (46, 46, 51) and p["order"] == 5                     # pme_params, ewald.jl:479-482
    assert case.inter_dict(np.float64)["ewald_alpha"] == pytest.approx(2.6283, abs(1e-4))
    assert orc.pme_mesh([0.5, 0.5, 0.5], 2.6283) == (6, 6, 6)                # max(s, 6)


@pytest.mark.parametrize("exact,ftol,etol", [(True, 1e-7, 1e-5), (False, 1e-3, 0.2)])           # test/protein.jl:267, 274
def test_all_pme_forces_and_energy_vs_openmm(exact, ftol, etol):
    d = G.data()
    case = G.case("ewald", np.float64, bonded=True, approx_erfc=not exact, pme=True)
    o = case.oracle(np.float64)
    nl = o.neighbors("cell", nthreads=8)
    f = o.forces(nl, nthreads=8, pairwise=True, specific=True, general=True)
    key = "all_pme_exact" if exact else "all_pme"
    assert np.linalg.norm(f - d[f"openmm_forces_{key}"], axis=1).max() < ftol
    e = o.potential_energy(nl, pairwise=True, specific=True, general=True) + G.lj_dispersion_correction(d)
    assert abs(e - float(d[f"openmm_energy_{key}"])) < etol


def test_pme_reciprocal_forces_sum_to_zero_and_energy_is_translation_invariant():
    case = G.case("ewald", np.float64, bonded=False, lj=False, pme=True)
    o = case.oracle(np.float64)
    f = o.forces(None, pairwise=False, specific=False, general=True)
    assert np.abs(f.sum(axis=0)).max() < 2e-2 * np.abs(f).max()            # mesh forces conserve momentum only to PME accuracy
    e0 = o.potential_energy(None, pairwise=False, general=True)
    o2 = case.oracle(np.float64, coords=case.coords + np.array([0.123, -0.4, 0.05]))
    o2.wrap()
    e1 = o2.potential_energy(None, pairwise=False, general=True)
    assert abs(e1 - e0) < 5e-4 * abs(e0)                                     # mesh discretisation only


def test_threaded_reciprocal_space_is_the_serial_one():
    """oracle/pme.h with threads (the cpu_baseline leg of bench.py: B-splines, interpolation, transforms and convolution over all threads,
    the spreading into min(n_threads, 4) private meshes as ewald.jl:888, 632-646) against the serial path: only the order of the mesh sums differs"""
    case = G.case("ewald", np.float64, bonded=False, lj=False, pme=True)
    o = case.oracle(np.float64)
    f1 = o.forces(None, nthreads=1, pairwise=False, specific=False, general=True)
    for nt in (2, 3, 8):
        ft = o.forces(None, nthreads=nt, pairwise=False, specific=False, general=True)
        assert np.abs(ft - f1).max() < 1e-10 * np.abs(f1).max()
    o32 = case.oracle(np.float32)
    g1 = o32.forces(None, nthreads=1, pairwise=False, specific=False, general=True).astype(np.float64)
    g8 = o32.forces(None, nthreads=8, pairwise=False, specific=False, general=True).astype(np.float64)
    assert np.abs(g8 - g1).max() < 1e-4 * np.abs(g1).max()


def test_100_step_pme_trajectory_vs_openmm():
    """test/protein.jl:278-299: 100 velocity-Verlet steps of 0.5 fs with every interaction incl. PME"""
    d = G.data()
    case = G.case("ewald", np.float64, bonded=True, approx_erfc=False, pme=True)
    o = case.oracle(np.float64)
    assert o.kinetic_energy() == pytest.approx(65521.87288132431, rel=1e-9)                 # :284
    o.vv_run(100, 0.0005, remove_cm_every=1, nthreads=min(orc.hardware_threads(), 32), specific=True, general=True)
    xo = d["openmm_coordinates_100steps"]; box = case.box
    dx = o.coords - (xo - np.floor(xo / box) * box)
    dx -= np.round(dx / box) * box
    assert np.linalg.norm(dx, axis=1).max() < 1e-10                                           # :297
    assert np.linalg.norm(o.vel - d["openmm_velocities_100steps"], axis=1).max() < 1e-7       # :298


def test_specific_and_pme_virial_obey_the_scaling_identity():
    """tr W = −dE/dλ under x → λx, L → λL for the specific interactions (force.jl:991-1060) and for the PME reciprocal part with α and
    the mesh fixed (ewald.jl:701-723, 925-927); both tensors symmetric."""
    from tests import systems as S
    case = G.case(None, np.float64, lj=False, bonded=True)
    w = case.oracle(np.float64).virial(None, pairwise=False, specific=True)

    def e_spec(lam):
        c2 = G.case(None, np.float64, lj=False, bonded=True)
        c2.coords = case.coords * lam; c2.box = case.box * lam
        return c2.oracle(np.float64).potential_energy(None, pairwise=False, specific=True)

    h = 1e-7
    assert np.trace(w) == pytest.approx(-(e_spec(1 + h) - e_spec(1 - h)) / (2 * h), rel=1e-7)
    assert np.abs(w - w.T).max() < 1e-9 * np.abs(w).max()

    mk = lambda: S.charged_fluid(8, dict(kind="ewald", rc=0.9, tol=5e-4), dtype=np.float64, with_exceptions=False, r_list=1.0, pme=dict(order=5, mesh=(14, 14, 14)))
    cp = mk()
    wp = cp.oracle(np.float64).virial(None, pairwise=False, general=True)

    def e_pme(lam):
        c2 = mk()
        c2.coords = cp.coords * lam; c2.box = cp.box * lam
        return c2.oracle(np.float64).potential_energy(None, pairwise=False, general=True)

    h = 1e-6
    assert np.trace(wp) == pytest.approx(-(e_pme(1 + h) - e_pme(1 - h)) / (2 * h), rel=1e-7)
    assert np.abs(wp - wp.T).max() < 1e-10 * np.abs(wp).max()


def _charges_in_a_box(n=600, L=3.1, seed=11):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, L, (n, 3))
    q = rng.uniform(-1, 1, n); q -= q.mean()
    return x, q, L


def _pme_case(x, q, L, basis=None, mesh=(32, 32, 32)):
    from tests import systems as S
    tri = None if basis is None else dict(basis=np.asarray(basis, dtype=np.float64), approx_images=False)
    box = [L, L, L] if basis is None else list(np.diag(np.asarray(basis, dtype=np.float64)))
    return S.Case(x, box, coul=dict(kind="ewald", rc=1.0, tol=5e-4), r_list=1.2, charge=q, sigma=np.zeros(len(x)), eps=np.zeros(len(x)), mass=np.ones(len(x)),
                  pme=dict(order=5, mesh=mesh), triclinic=tri)


def test_triclinic_reciprocal_space_zero_tilt_is_the_cubic_one_and_a_sheared_cell_of_the_same_lattice_agrees():
    """PME on a TriclinicBoundary (recip_box = invert_box_vectors(boundary), spatial.jl:338-347, in ewald.jl:486, 688-694, 846-849).  No fixture of the
    reference covers it (test/gradients.jl only differentiates it), so it is pinned by two properties: a basis without tilt reproduces the cubic
    numbers bit for bit, and the cell (a, a + b, c) — another cell of the SAME lattice — gives the same reciprocal-space forces and energy up to
    the discretisation error of its differently oriented mesh."""
    x, q, L = _charges_in_a_box()
    cubic = _pme_case(x, q, L).oracle(np.float64)
    f0 = cubic.forces(None, pairwise=False, specific=False, general=True)
    e0 = cubic.potential_energy(None, pairwise=False, general=True)
    flat = _pme_case(x, q, L, basis=[[L, 0, 0], [0, L, 0], [0, 0, L]]).oracle(np.float64)
    assert np.array_equal(flat.forces(None, pairwise=False, specific=False, general=True), f0)
    assert flat.potential_energy(None, pairwise=False, general=True) == e0
    # a sheared cell's mesh lines are longer: compared on a 64³ mesh, where the cubic cell's own result has settled to 1e-4 of the largest force
    fine = _pme_case(x, q, L, mesh=(64, 64, 64)).oracle(np.float64)
    f0 = fine.forces(None, pairwise=False, specific=False, general=True)
    e0 = fine.potential_energy(None, pairwise=False, general=True)
    for basis in ([[L, 0, 0], [L, L, 0], [0, 0, L]], [[L, 0, 0], [0, L, 0], [L, -L, L]], [[L, 0, 0], [-L, L, 0], [L, L, L]]):
        sh = _pme_case(x, q, L, basis=basis, mesh=(64, 64, 64)).oracle(np.float64)
        f1 = sh.forces(None, pairwise=False, specific=False, general=True)
        e1 = sh.potential_energy(None, pairwise=False, general=True)
        assert np.abs(f1 - f0).max() < 5e-5 * np.abs(f0).max(), (basis, np.abs(f1 - f0).max() / np.abs(f0).max())
        assert abs(e1 - e0) < 0.1, (basis, e1 - e0)


def test_triclinic_reciprocal_virial_is_the_strain_derivative_of_the_energy():
    """The reciprocal-space virial on a sheared cell (recip_conv_inner!, ewald.jl:701-723, with m · recip_box) has no fixture in the reference either; it is pinned by
    what a virial is: W_ab = −∂E/∂ε_ab under the homogeneous strain x → (1 + ε) x of coordinates and cell.  Central differences of the oracle's own reciprocal energy
    (self term constant, net-charge term ∝ 1/V: its share charge_E · I is part of the tensor, :925-927) on a 48³ mesh, the three diagonal strains and the three shears."""
    x, q, L = _charges_in_a_box(n=300, L=2.6, seed=3)
    q = q + 0.02                                                    # a net charge: the charge_E term takes part
    basis = np.array([[L, 0, 0], [0.35 * L, L, 0], [-0.2 * L, 0.3 * L, L]])

    def energy_and_virial(strain):
        F = np.eye(3) + strain                                      # x' = F x (row vectors: x @ F.T)
        o = _pme_case(x @ F.T, q, L, basis=basis @ F.T, mesh=(48, 48, 48)).oracle(np.float64)
        return o.potential_energy(None, pairwise=False, general=True), o.virial(None, pairwise=False, specific=False, general=True)

    e0, w = energy_and_virial(np.zeros((3, 3)))
    h = 2e-5
    for a, b in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)):      # (upper-triangular shears keep v1 along x and v2 in the xy plane)
        s = np.zeros((3, 3)); s[a, b] = h
        de = (energy_and_virial(s)[0] - energy_and_virial(-s)[0]) / (2 * h)
        # the B-spline weights do not change under a homogeneous strain, so the mesh energy's derivative IS the tensor: agreement at the level of the difference quotient
        assert abs(-de - w[a, b]) < 1e-6 * np.abs(w).max(), ((a, b), -de, w[a, b])
    assert np.abs(w - w.T).max() < 1e-9 * np.abs(w).max()


def test_triclinic_reciprocal_forces_are_the_gradient_of_the_energy():
    """interpolate_force_inner! with a triclinic recip_box (ewald.jl:846-849): the smooth-PME force is the exact gradient of the mesh energy, so central differences of
    the oracle's reciprocal energy must give it back — on a sheared cell, for a handful of atoms and all three components"""
    x, q, L = _charges_in_a_box(n=300, L=2.6, seed=5)
    basis = np.array([[L, 0, 0], [0.35 * L, L, 0], [-0.2 * L, 0.3 * L, L]])
    f = _pme_case(x, q, L, basis=basis, mesh=(40, 42, 45)).oracle(np.float64).forces(None, pairwise=False, specific=False, general=True)
    h = 1e-5
    for i in (0, 17, 123, 299):
        for d in range(3):
            xp, xm = x.copy(), x.copy()
            xp[i, d] += h; xm[i, d] -= h
            ep = _pme_case(xp, q, L, basis=basis, mesh=(40, 42, 45)).oracle(np.float64).potential_energy(None, pairwise=False, general=True)
            em = _pme_case(xm, q, L, basis=basis, mesh=(40, 42, 45)).oracle(np.float64).potential_energy(None, pairwise=False, general=True)
            assert abs(-(ep - em) / (2 * h) - f[i, d]) < 1e-6 * np.abs(f).max(), (i, d, -(ep - em) / (2 * h), f[i, d])


def test_triclinic_ewald_total_energy_does_not_depend_on_the_splitting():
    """direct space (minimum image on the sheared cell, erfc) + reciprocal space (triclinic recip_box) + self term: the Ewald sum's value must not depend on how α
    splits it between the two — two cutoffs give two α (α = √(−ln 2 tol) / r_c, ewald.jl:368), the totals agree to the sums' own tolerance"""
    x, q, L = _charges_in_a_box(n=250, L=2.8, seed=9)
    basis = np.array([[L, 0, 0], [0.3 * L, L, 0], [0.2 * L, -0.25 * L, L]])
    e = []
    for rc in (0.8, 1.15):
        from tests import systems as S
        case = S.Case(x, list(np.diag(basis)), coul=dict(kind="ewald", rc=rc, tol=1e-6, approx=False), r_list=rc + 0.05, charge=q, sigma=np.zeros(len(x)), eps=np.zeros(len(x)),
                      mass=np.ones(len(x)), pme=dict(order=6, mesh=(72, 72, 72)), triclinic=dict(basis=basis, approx_images=False))
        o = case.oracle(np.float64)
        e.append(o.potential_energy(o.neighbors("brute"), pairwise=True, general=True))
    assert abs(e[0] - e[1]) < 2e-5 * abs(e[0]), e

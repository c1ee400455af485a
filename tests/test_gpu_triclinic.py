"""TriclinicBoundary on the MI355X (mhip_set_triclinic): the reference's own GPU test (test/gpu_consistency.jl:287-337) and its
free-flight test (test/basic.jl:233-260) through the C ABI, neighbour lists, dynamics and the bonded terms against the oracle."""
import numpy as np
import pytest

from tests import systems as S
from tests.test_oracle_triclinic import BASIS, tri_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("approx", [True, False])
@pytest.mark.parametrize("dtype,rtol,atol", [(np.float64, 1e-8, 1e-10), (np.float32, 2e-4, 1e-3)])
def test_triclinic_forces_energy_and_list_match_oracle(pkg, dtype, rtol, atol, approx):
    """test/gpu_consistency.jl:287-337: 50 LJ atoms (σ 0.3, ϵ 1) in the cell (2,0,0), (0.1,2,0), (0.2,0.3,2), cutoff 0.8; GPU forces and
    energy against the CPU path at rtol 1e-8 / atol 1e-10 (fp64)"""
    case = tri_case(50, dtype, approx)
    o = case.oracle(np.float64)
    nl = o.neighbors("brute")
    f_ref, e_ref = o.forces(nl), o.potential_energy(nl)
    s = case.system(pkg, dtype)
    f = pkg.forces(s).astype(np.float64)
    assert np.all(np.abs(f - f_ref) <= rtol * np.abs(f_ref) + atol * max(1.0, np.abs(f_ref).max() if dtype == np.float32 else 1.0))
    assert pkg.potential_energy(s) == pytest.approx(e_ref, rel=rtol, abs=atol)
    got = pkg.find_neighbors(s)
    ref = case.oracle(dtype).neighbors("brute")
    assert all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(got.i, got.j, got.special), S.sorted_pairs(*ref)))   # bit-exact pair set


def dense_case(seed=11):
    """343 atoms on a jittered lattice in fractional coordinates of a strongly sheared cell: crosses every face, no overlaps"""
    rng = np.random.default_rng(seed)
    n_side = 7
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    basis = np.array([[2.4, 0.0, 0.0], [0.5, 2.3, 0.0], [-0.4, 0.6, 2.2]])
    x = ((g + 0.5) / n_side + rng.uniform(-0.02, 0.02, g.shape)) @ basis
    n = len(x)
    v = rng.normal(size=(n, 3)) * 0.3
    return basis, S.Case(x, np.diag(basis), lj=dict(cutoff=("shifted_force", 0.9)), r_list=1.0, rebuild_every=5, velocities=v, sigma=np.full(n, 0.3),
                         eps=np.full(n, 0.8), mass=np.full(n, 12.0), triclinic=dict(basis=basis), name="tri_dense")


def test_triclinic_dense_fluid_trajectory_matches_oracle(pkg):
    """a denser system that crosses every face of the cell: 40 velocity-Verlet steps with CM removal, then 20 Langevin steps, fp64
    against the oracle; coordinates stay wrapped into the cell"""
    basis, case = dense_case()
    o = case.oracle(np.float64)
    o.vv_run(40, 0.002, remove_cm_every=1)
    s = case.system(pkg, np.float64)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), 40)
    d = np.linalg.solve(basis.T, (s.coords - o.coords).T).T
    d -= np.round(d)
    assert np.abs(d @ basis).max() < 1e-9 and np.abs(s.velocities - o.vel).max() < 1e-8
    frac = np.linalg.solve(basis.T, s.coords.T).T
    assert frac.min() > -1e-12 and frac.max() < 1 + 1e-12                   # wrapped into the cell
    rng = np.random.default_rng(4)
    key, ctr1 = (int(rng.integers(0, 2 ** 64, dtype=np.uint64)) for _ in range(2))
    o.langevin_run(20, 0.002, 8.314462618e-3 * 200.0, 2.0, key=key, ctr1=ctr1, first_step=40, remove_cm_every=1)
    pkg.simulate(s, pkg.Langevin(dt=0.002, temperature=200.0, friction=2.0), 20, init_step=40, rng=4)
    d = np.linalg.solve(basis.T, (s.coords - o.coords).T).T
    d -= np.round(d)
    assert np.abs(d @ basis).max() < 1e-9


def test_triclinic_free_flight(pkg):
    """test/basic.jl:233-260 on the device: no forces, 1000 steps — velocities unchanged, coordinates wrapped, displacements = v·t"""
    basis = np.array([[2.2, 0.0, 0.0], [1.0, 1.7320508075688772, 0.0], [1.3788800, 0.5399122, 1.0233204]])
    rng = np.random.default_rng(5)
    n = 1000
    x = rng.uniform(0, 1, (n, 3)) @ basis
    v = rng.normal(size=(n, 3)) * np.sqrt(8.314462618e-3 * 100.0)
    case = S.Case(x, np.diag(basis), lj=dict(cutoff=("distance", 0.4)), r_list=0.45, velocities=v, sigma=np.full(n, 0.3), eps=np.zeros(n), mass=np.ones(n),
                  triclinic=dict(basis=basis))
    s = case.system(pkg, np.float64)
    s.push_state(); s.pull_state()
    prev = s.coords.copy()
    for k in range(10):
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.002, remove_CM_motion=0), 100, init_step=100 * k)
        assert np.array_equal(pkg.wrap_coords(s.coords, s.boundary), s.coords)
        d = np.linalg.solve(basis.T, (s.coords - prev - 0.2 * v).T).T
        assert np.abs(d - np.round(d)).max() < 1e-9
        prev = s.coords.copy()
    assert np.allclose(s.velocities, v, rtol=1e-12, atol=0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_triclinic_neighbour_list_of_1000_atoms_is_the_brute_force_set(pkg, dtype):
    """test/basic.jl:544-577: 1000 atoms placed in the cell (2,0,0), (0.7,1.8,0), (0.5,0.3,1.6), cutoff 0.6 nm — the list equals the pairs
    with norm(vector(ci, cj, boundary)) <= cutoff, here as an exact set comparison in the same precision"""
    basis = np.array([[2.0, 0.0, 0.0], [0.7, 1.8, 0.0], [0.5, 0.3, 1.6]])
    rng = np.random.default_rng(21)
    n = 1000
    x = (rng.uniform(0, 1, (n, 3)) @ basis).astype(dtype).astype(np.float64)
    case = S.Case(x, np.diag(basis), lj=dict(cutoff=("distance", 0.6)), r_list=0.6, velocities=np.zeros((n, 3)), sigma=np.full(n, 0.05), eps=np.full(n, 0.1),
                  mass=np.ones(n), triclinic=dict(basis=basis), name="tri1000")
    ref = case.oracle(dtype).neighbors("brute")
    s = case.system(pkg, dtype)
    got = pkg.find_neighbors(s)
    assert got.n == len(ref[0]) and got.n > 20000
    assert all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(got.i, got.j, got.special), S.sorted_pairs(*ref)))


def sheared_fluid(dtype, n_side=20, seed=17, jitter=0.008):
    """n_side³ argon-like atoms on a jittered lattice in the fractional coordinates of a sheared cell whose perpendicular heights
    (≈ 7 nm at 8000 atoms) hold 11 cells of r_list / 2 on every axis and leave every 64-atom block's neighbourhood well inside half a
    height: the cell-grid form of the triclinic search with block-local coordinates, not the one-cell form"""
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    basis = np.array([[6.0, 0.0, 0.0], [1.3, 5.8, 0.0], [-0.9, 1.5, 5.6]]) * (n_side / 16.0)
    x = (((g + 0.5) / n_side + rng.uniform(-jitter, jitter, g.shape)) @ basis).astype(dtype).astype(np.float64)
    n = len(x)
    v = (rng.normal(size=(n, 3)) * 0.13).astype(dtype).astype(np.float64)
    v -= v.mean(axis=0)
    return basis, S.Case(x, np.diag(basis), lj=dict(cutoff=("distance", 1.0)), r_list=1.2, rebuild_every=10, velocities=v, sigma=np.full(n, 0.34),
                         eps=np.full(n, 0.997), mass=np.full(n, 39.948), triclinic=dict(basis=basis), name=f"tri_fluid{n}")


@pytest.mark.parametrize("approx", [True, False])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_triclinic_cell_grid_list_and_forces(pkg, dtype, approx):
    """a triclinic box large enough for a cell grid (spatial.jl:528-551 for the images, neighbors.jl:390-423 for the predicate): the
    pair set equals the brute-force set of the same precision, forces and energy match the fp64 oracle, and the blocks really saw
    only their neighbourhood (tiles smaller than the system, block-local coordinates instead of the in-loop minimum image)"""
    basis, case = sheared_fluid(dtype)
    case.triclinic["approx_images"] = approx
    ref = case.oracle(dtype).neighbors("brute", nthreads=8)
    s = case.system(pkg, dtype)
    got = pkg.find_neighbors(s)
    assert got.n == len(ref[0])
    assert all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(got.i, got.j, got.special), S.sorted_pairs(*ref)))
    st = s.stats()
    assert st["minimg_mode"] == 0 and st["max_tile_atoms"] < 3 * case.n // 4, st
    o = case.oracle(np.float64)
    nl = o.neighbors("brute", nthreads=8)
    f_ref, e_ref = o.forces(nl, nthreads=4), o.potential_energy(nl)
    f = pkg.forces(s).astype(np.float64)
    if dtype == np.float64:
        assert np.abs(f - f_ref).max() < 1e-8 * np.abs(f_ref).max()
        assert pkg.potential_energy(s) == pytest.approx(e_ref, rel=1e-10)
    else:
        scale, jump = o.force_scale(nl)      # (jump: the force step of a pair within 2e-6 of the hard cutoff, where fp32 may flip r <= rc — tests/systems.py fp32_force_tolerance)
        assert np.all(np.linalg.norm(f - f_ref, axis=1) <= 6e-5 * scale + 1.01 * jump + 1e-4)
        assert pkg.potential_energy(s) == pytest.approx(e_ref, rel=3e-5)


def test_triclinic_single_list_with_exceptions_beyond_64_atom_blocks(pkg, monkeypatch):
    """the search variant with exact band decisions and exception lookups (k_build<T, true, false, true>) lost the lists of every i-wave but the first beyond
    64-atom blocks and was detoured there — on a triclinic cell grid, which has no transposed search, by cutting the blocks to 64 atoms.  Round 6 bisected the defect
    to private arrays in scratch that a later commit had already removed (engine.hip rebuild_impl) and took the detour out: 42 875 atoms in a sheared cell now search in
    128-atom blocks.  Excluded and special pairs, the single exact list (no outer margin): pair SET and special flags against the brute-force oracle of the same
    precision, forces against the fp64 oracle."""
    monkeypatch.setenv("MOLLYHIP_OUTER_MARGIN_PM", "0")
    # (the jitter is fractional: scaled so that it stays the ±0.06 nm of the 8 000-atom cell — at ±0.1 nm atoms overlap, forces reach 2·10⁵ kJ mol⁻¹ nm⁻¹ ∝ r⁻¹³ and the
    # fp32 rounding of a coordinate in a 13 nm cell, 1e-6 nm, is 1e-4 of such a force: tools/micro/tri_xl_check.py)
    basis, case = sheared_fluid(np.float32, n_side=35, jitter=0.008 * 20 / 35)
    idx = np.arange(case.n - 2)
    case.excluded = np.stack([idx[idx % 3 == 0], idx[idx % 3 == 0] + 1], 1)
    case.special = np.stack([idx[idx % 3 == 1], idx[idx % 3 == 1] + 2], 1)
    case.lj = dict(cutoff=("distance", 1.0), weight_special=0.5)
    ref = case.oracle(np.float32).neighbors("brute", nthreads=16)
    s = case.system(pkg, np.float32)
    got = pkg.find_neighbors(s)
    assert got.n == len(ref[0])
    assert all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(got.i, got.j, got.special), S.sorted_pairs(*ref)))
    assert s.stats()["block_atoms"] == 128
    o = case.oracle(np.float64)
    nl = o.neighbors("brute", nthreads=16)
    f_ref = o.forces(nl, nthreads=8)
    scale, jump = o.force_scale(nl)      # (jump: the force step of a pair within 2e-6 of the hard cutoff, where fp32 may flip r <= rc — 0.037 kJ/mol/nm for argon at 1 nm)
    f = pkg.forces(s).astype(np.float64)
    # 7e-5 where the 8 000-atom cell above holds 6e-5: in this 13 nm cell an fp32 coordinate's ulp is twice that of the 7.5 nm cell.  Measured 5.7e-5 at the worst atom
    # (tools/micro/tri_xl_check.py); before round 6 7.6e-5 — tile coordinates went to fractional coordinates and back, an ulp of the CELL per atom, where local_xyz_t
    # now takes Cartesian differences against the centre and whole lattice vectors
    assert np.all(np.linalg.norm(f - f_ref, axis=1) <= 7e-5 * scale + 1.01 * jump + 1e-4)


def test_triclinic_cell_grid_trajectory(pkg):
    """60 velocity-Verlet steps (six list rebuilds, atoms crossing every face) of the sheared fluid against the oracle, fp64"""
    basis, case = sheared_fluid(np.float64, n_side=20)
    o = case.oracle(np.float64)
    o.vv_run(60, 0.002, remove_cm_every=1, nthreads=8)
    s = case.system(pkg, np.float64)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), 60)
    d = np.linalg.solve(basis.T, (s.coords - o.coords).T).T
    d -= np.round(d)
    assert np.abs(d @ basis).max() < 1e-9 and np.abs(s.velocities - o.vel).max() < 1e-8
    st = s.stats()
    assert st["minimg_mode"] == 0
    # the dual pair list works in the sheared cell too: searches with r_list + margin when the displacement bounds (nearest image) ask
    # for them — the lattice is melting: three in these 60 steps —, the inner list pruned from the outer one in between; the fixed
    # cadence of the one-cell form searches at the start and at each of the six rebuild steps
    assert st["n_outer_builds"] <= 4 and st["n_filter_passes"] >= 1, st
    # … and what the engine hands out after those 60 steps is still exactly the reference's list of the coordinates reached
    got = pkg.find_neighbors(s)
    ref = case.oracle(np.float64, coords=s.coords).neighbors("brute", nthreads=8)
    assert got.n == len(ref[0])
    assert all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(got.i, got.j, got.special), S.sorted_pairs(*ref)))


def test_triclinic_fp32_fluid_trajectory_with_pruned_lists(pkg):
    """the same fluid in fp32, 120 steps: prunes of the inner list on the way, trajectory within the fp32 bar of the reference's own
    test (mean deviation below 5e-4 nm after 100 steps, test/simulation.jl:625)"""
    basis, case = sheared_fluid(np.float32, n_side=20)
    o = case.oracle(np.float64)
    o.vv_run(120, 0.002, remove_cm_every=1, nthreads=8)
    s = case.system(pkg, np.float32)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), 120)
    d = np.linalg.solve(basis.T, (s.coords.astype(np.float64) - o.coords).T).T
    d -= np.round(d)
    assert np.linalg.norm(d @ basis, axis=1).mean() < 5e-4
    st = s.stats()
    assert st["n_outer_builds"] <= 7 and st["n_filter_passes"] >= 2, st      # (13 searches at the fixed cadence)


def test_triclinic_bonded_terms(pkg):
    """bonds and angles across the faces of the cell take the same minimum image (force.jl:991-1060 with vector(…, boundary))"""
    case = tri_case(48, np.float64, True, seed=8, spread=1.9)
    i = np.arange(0, 48, 3)
    case.bonds = dict(i=i, j=i + 1, k=np.full(len(i), 2000.0), r0=np.full(len(i), 0.35))
    case.angles = dict(i=i, j=i + 1, k=i + 2, kth=np.full(len(i), 300.0), th0=np.full(len(i), 1.9))
    o = case.oracle(np.float64)
    f_ref = o.forces(None, pairwise=False, specific=True)
    s = case.system(pkg, np.float64)
    f = pkg.forces(s, pairwise=False)
    assert np.abs(f - f_ref).max() < 1e-9 * np.abs(f_ref).max()
    assert pkg.potential_energy(s, pairwise=False) == pytest.approx(o.potential_energy(None, pairwise=False, specific=True), rel=1e-11)


def test_triclinic_virial_and_pressure(pkg):
    """virial(sys) = Σ dr ⊗ f with the boundary's minimum image (force.jl:848-852, 991-1060); the volume of the cell is v1.x·v2.y·v3.z"""
    basis, case = dense_case(seed=3)
    i = np.arange(0, 342, 3)
    case.bonds = dict(i=i, j=i + 1, k=np.full(len(i), 500.0), r0=np.full(len(i), 0.33))
    o = case.oracle(np.float64)
    nl = o.neighbors("brute")
    w_ref = o.virial(nl, pairwise=True, specific=True)
    s = case.system(pkg, np.float64)
    w = pkg.virial(s)
    assert np.abs(w - w_ref).max() < 1e-9 * np.abs(w_ref).max()
    k = 0.5 * np.einsum("i,ia,ib->ab", case.mass, case.velocities, case.velocities)
    assert np.abs(pkg.pressure(s) - (2 * k + w_ref) / np.prod(np.diag(basis))).max() < 1e-9 * np.abs(w_ref).max() / np.prod(np.diag(basis))


def test_triclinic_constructor_refusals_and_pme_on_a_sheared_cell(pkg):
    with pytest.raises(ValueError):
        pkg.TriclinicBoundary((2.0, 1.0, 0.0), (1.0, 2.0, 0.0), (1.0, 1.0, 2.0))       # test/basic.jl:202-206
    case = S.charged_fluid(6, dict(kind="ewald", rc=0.9), dtype=np.float64, with_exceptions=False, pme=dict(order=5))
    case.triclinic = dict(basis=np.diag(case.box) + np.array([[0, 0, 0], [0.1, 0, 0], [0, 0, 0]]))
    s = case.system(pkg, np.float64)
    pkg.forces(s); pkg.virial(s)                                                       # PME on a sheared cell works (tests/test_gpu_pme.py: forces, energy, virial)


@pytest.mark.parametrize("approx", [True, False])
@pytest.mark.parametrize("kind", ["rf", "ewald"])
def test_triclinic_fp32_charged_per_atom_lj_small_cell_is_finite_and_matches_oracle(pkg, kind, approx):
    """fp32, per-atom σ / ϵ, CoulombReactionField (and Ewald direct space) in the reference's small sheared cell: every block takes the exact
    in-loop minimum image and the two-partner packed loop, whose padded slots are pushed out to r² = 1e12 — with 1e30 the reaction-field
    branch formed r³ = +inf and inf·0 made every force NaN.  Forces finite and inside the fp32 bar of the fp64 oracle."""
    rng = np.random.default_rng(9)
    n = 90
    g = np.stack(np.meshgrid(*[np.arange(5)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n].astype(np.float64)
    x = (((g + 0.5) / 5 + rng.uniform(-0.03, 0.03, g.shape)) @ BASIS).astype(np.float32).astype(np.float64)
    q = rng.uniform(-0.8, 0.8, n); q -= q.mean()
    coul = dict(kind="rf", rc=0.8, eps_rf=78.3) if kind == "rf" else dict(kind="ewald", rc=0.8, tol=5e-4)
    case = S.Case(x, np.diag(BASIS), lj=dict(cutoff=("distance", 0.8)), coul=coul, r_list=0.9, velocities=np.zeros((n, 3)), charge=q,
                  sigma=rng.uniform(0.2, 0.32, n), eps=rng.uniform(0.3, 1.0, n), mass=np.full(n, 12.0),
                  triclinic=dict(basis=BASIS, approx_images=approx), name="tri_rf32")
    tol, o, nl = S.fp32_force_tolerance(case)
    f_ref = o.forces(nl)
    s = case.system(pkg, np.float32)
    f = pkg.forces(s).astype(np.float64)
    assert np.isfinite(f).all()
    err = np.linalg.norm(f - f_ref, axis=1)
    S.fp32_check(err, tol)
    assert S.rel_rms(err, f_ref) < 2e-5

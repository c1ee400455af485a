"""Edge cases of the hot path through the C ABI: the smallest systems, ragged and mostly empty grids, atoms on cell and box faces,
tiles that have to be cut into LDS segments — each against the oracle (pair set bit-exact, forces at the fp64 / fp32 bars)."""
import numpy as np
import pytest

from tests import systems as S
from tests.test_gpu_parity import assert_same_neighbors

pytestmark = pytest.mark.gpu


def _check_forces(pkg, case, dtype, rel64=1e-9):
    s = case.system(pkg, dtype)
    f = pkg.forces(s)
    if np.dtype(dtype) == np.float64:
        o = case.oracle(np.float64)
        nl = o.neighbors("cell") if np.isfinite(case.r_list) else None
        fo = o.forces(nl)
        assert np.abs(f - fo).max() <= rel64 * max(np.abs(fo).max(), 1.0)
        assert pkg.potential_energy(s) == pytest.approx(o.potential_energy(nl), rel=1e-10, abs=1e-10)
    else:
        tol, o, nl = S.fp32_force_tolerance(case)
        fo = o.forces(nl)
        S.fp32_check(np.linalg.norm(f.astype(np.float64) - fo, axis=1), tol)
    return s, f


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_one_atom_and_two_atoms(pkg, dtype):
    one = S.Case([[1.0, 1.0, 1.0]], 4.0, lj=dict(cutoff=("distance", 1.0)), r_list=1.2, velocities=[[0.3, -0.2, 0.1]],
                 sigma=[0.3], eps=[0.5], mass=[10.0])
    s = one.system(pkg, dtype)
    assert np.all(pkg.forces(s) == 0) and pkg.potential_energy(s) == 0
    nl = pkg.find_neighbors(s)
    assert nl.n == 0
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002, remove_CM_motion=0), 25)       # free flight through two list refreshes
    assert np.allclose(s.coords, np.array([[1.0, 1.0, 1.0]]) + 25 * 0.002 * np.array([[0.3, -0.2, 0.1]]), atol=2e-6)
    for d, n_pairs in ((0.35, 1), (1.1, 1), (1.3, 0)):                           # inside the cutoff, inside the list only, outside both
        xy = np.array([[0.2, 2.0, 2.0], [0.2 - d + 4.0, 2.0, 2.0]]).astype(dtype).astype(np.float64)    # every precision sees the same inputs
        two = S.Case(xy, 4.0, lj=dict(cutoff=("distance", 1.0)), r_list=1.2,   # across the periodic face
                     sigma=[0.3, 0.32], eps=[0.5, 0.4], mass=[10.0, 12.0])
        s, f = _check_forces(pkg, two, dtype)
        assert pkg.find_neighbors(s).n == n_pairs
        assert np.array_equal(f[0], -f[1])                                       # one pair, evaluated from both ends with the same arithmetic
        assert (np.abs(f).max() > 0) == (d < 1.0)
    nolist = S.Case([[1.0, 1.0, 1.0], [1.4, 1.0, 1.0]], 4.0, lj=dict(cutoff=("none",)), sigma=[0.3, 0.3], eps=[0.5, 0.5], mass=[10.0, 10.0])
    _check_forces(pkg, nolist, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_droplet_in_a_mostly_empty_box(pkg, dtype):
    """2 999 atoms (not a multiple of any block size) packed into one corner region of a 24 nm box: almost every cell of the grid is
    empty, the last block is ragged, the droplet straddles three periodic faces"""
    rng = np.random.default_rng(11)
    n_side, sp = 15, 0.33
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:2999].astype(np.float64)
    box = 24.0
    x = (g - 7.0) * sp + rng.uniform(-0.03, 0.03, g.shape)                       # centred on the origin: wraps to all eight corners
    x = x - np.floor(x / box) * box
    x = x.astype(dtype).astype(np.float64); x = np.where(x >= box, 0.0, x)
    v = rng.normal(size=x.shape) * 0.15; v -= v.mean(0)
    case = S.Case(x, box, lj=dict(cutoff=("distance", 1.0)), r_list=1.2, velocities=v.astype(dtype).astype(np.float64),
                  sigma=np.full(len(x), 0.3), eps=np.full(len(x), 0.4), mass=np.full(len(x), 20.0), name="droplet")
    assert_same_neighbors(pkg, case, dtype)
    _check_forces(pkg, case, dtype)
    if np.dtype(dtype) == np.float64:
        o = case.oracle(np.float64)
        o.vv_run(30, 0.002, remove_cm_every=1)
        s = case.system(pkg, np.float64)
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), 30)
        d = s.coords - o.coords; d -= np.round(d / box) * box
        assert np.abs(d).max() < 1e-9


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_atoms_on_cell_and_box_faces(pkg, dtype):
    """coordinates that are exact multiples of the cell size (0, L/nc, 2L/nc, …) and the largest value below L: every atom sits on a
    face of the search grid, pair distances are exact multiples too, so r == r_list and r == rc happen exactly"""
    box, m = 7.2, 12                                                             # 0.6 nm lattice: r_list 1.2 = two spacings exactly
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    x = (g * (box / m)).astype(dtype).astype(np.float64)
    face = np.flatnonzero(g[:, 0] == 0)[::3]
    x[face, 0] = np.float64(np.nextafter(dtype(box), dtype(0)))                  # a third of the x = 0 face moves to just below x = L
    case = S.Case(x, float(dtype(box)), lj=dict(cutoff=("distance", 1.0)), r_list=1.2, sigma=np.full(len(x), 0.3), eps=np.full(len(x), 0.4),
                  mass=np.full(len(x), 20.0), name="faces")
    assert_same_neighbors(pkg, case, dtype)                                      # `<=` at exactly r_list, decided with the reference's arithmetic
    _check_forces(pkg, case, dtype)


@pytest.mark.parametrize("maker,dtype", [("charged", np.float64), ("charged", np.float32), ("lj", np.float32), ("lj", np.float64)])
def test_tiles_cut_into_lds_segments(pkg, monkeypatch, maker, dtype):
    """a tile that does not fit the LDS is walked in segments (`SEG` variants of the pair kernel): forced here by a small LDS budget,
    with and without the pruning pass of the dual list; forces and energy must not depend on the cut"""
    case = S.charged_fluid(12, dict(kind="ewald", rc=1.0), dtype=dtype) if maker == "charged" else S.lj_fluid(16, dtype=dtype)
    s0, f0 = _check_forces(pkg, case, dtype)
    e0 = pkg.potential_energy(s0)
    monkeypatch.setenv("MOLLYHIP_LDS_BUDGET_KB", "17")                          # 16 KiB go to the j-split reduction: a few dozen tile atoms per segment
    s1, f1 = _check_forces(pkg, case, dtype)
    st = s1.stats()
    assert st["tile_segments"] >= 8 and s0.stats()["tile_segments"] == 1         # really segmented
    scale = max(np.abs(f0).max(), 1.0)
    assert np.abs(f1.astype(np.float64) - f0.astype(np.float64)).max() <= (1e-10 if np.dtype(dtype) == np.float64 else 2e-5) * scale
    assert pkg.potential_energy(s1) == pytest.approx(e0, rel=1e-10 if np.dtype(dtype) == np.float64 else 2e-6)
    pkg.simulate(s1, pkg.VelocityVerlet(dt=0.001), 25)                           # prunes and refreshes through the segmented kernels
    monkeypatch.delenv("MOLLYHIP_LDS_BUDGET_KB")
    s2 = case.system(pkg, dtype)
    pkg.simulate(s2, pkg.VelocityVerlet(dt=0.001), 25)
    d = s1.coords.astype(np.float64) - s2.coords.astype(np.float64); d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < (1e-10 if np.dtype(dtype) == np.float64 else 2e-5)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_random_gas_of_the_reference_tile_list_test(pkg, dtype):
    """test/gpu_consistency.jl:159-216 ("GPU tile lists"): 100 LJ atoms (σ 0.3, ϵ 1, mass 10) at uniformly random positions in a 10 nm box, cutoff 3 nm — GPU forces and
    energy over the GPUNeighborFinder against the CPU's all-pairs loop with the same cutoff (rtol 1e-8 / atol 1e-10 there).  Random positions put some pairs far
    inside σ (forces of 1e10 and more): the relative bar is what holds."""
    rng = np.random.default_rng(42)
    n = 100
    x = (rng.random((n, 3)) * 10.0).astype(dtype).astype(np.float64)
    case = S.Case(x, 10.0, lj=dict(cutoff=("distance", 3.0)), r_list=3.0, rebuild_every=10, sigma=np.full(n, 0.3), eps=np.full(n, 1.0), mass=np.full(n, 10.0))
    o = case.oracle(np.float64)
    nl = o.neighbors("brute")
    f_ref, e_ref = o.forces(nl), o.potential_energy(nl)
    s = case.system(pkg, dtype)
    f = pkg.forces(s).astype(np.float64)
    got = pkg.find_neighbors(s)
    if dtype == np.float64:
        assert got.n == len(nl[0]) and all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(got.i, got.j, got.special), S.sorted_pairs(*nl)))
        scale, _ = o.force_scale(nl)                                           # Σ_j‖f_ij‖: what the round-off of a sum in another order is relative to
        assert np.all(np.linalg.norm(f - f_ref, axis=1) <= 1e-8 * np.linalg.norm(f_ref, axis=1) + 1e-10 + 1e-13 * scale)
        assert pkg.potential_energy(s) == pytest.approx(e_ref, rel=1e-10, abs=1e-10)
    else:
        scale, jump = o.force_scale(nl)
        assert np.all(np.linalg.norm(f - f_ref, axis=1) <= 4e-5 * scale + 1.01 * jump + 1e-6)
        assert pkg.potential_energy(s) == pytest.approx(e_ref, rel=2e-5)


def test_cutoff_beyond_the_box_needs_no_tile_capacity(pkg):
    """test/gpu_consistency.jl:218-285: the same gas with a 20 nm cutoff in the 10 nm box — there the interacting-tile buffers overflow a deliberately tiny capacity and
    the call must throw (ext/MollyCUDAExt.jl:733-739).  This engine has no fixed tile capacity to overflow (capacities grow and the search is redone, csrc/engine.hip
    rebuild_impl), so the same system must simply come out right: every pair interacts once, through its minimum image."""
    rng = np.random.default_rng(42)
    n = 100
    x = rng.random((n, 3)) * 10.0
    case = S.Case(x, 10.0, lj=dict(cutoff=("distance", 20.0)), r_list=20.0, rebuild_every=10, sigma=np.full(n, 0.3), eps=np.full(n, 1.0), mass=np.full(n, 10.0))
    o = case.oracle(np.float64)
    nl = o.neighbors("brute")
    assert len(nl[0]) == n * (n - 1) // 2
    s = case.system(pkg, np.float64)
    f, f_ref = pkg.forces(s), o.forces(nl)
    scale, _ = o.force_scale(nl)
    assert np.all(np.linalg.norm(f - f_ref, axis=1) <= 1e-8 * np.linalg.norm(f_ref, axis=1) + 1e-10 + 1e-13 * scale)
    assert pkg.potential_energy(s) == pytest.approx(o.potential_energy(nl), rel=1e-10)
    assert pkg.find_neighbors(s).n == n * (n - 1) // 2

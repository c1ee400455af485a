"""TriclinicBoundary in the oracle (spatial.jl:131-220, 528-551, 588-602): the reference's own known answers (test/basic.jl) where it has
them, and the geometric properties that define the functions (CPU only)."""
import numpy as np
import pytest

from tests import systems as S

BASIS = np.array([[2.0, 0.0, 0.0], [0.1, 2.0, 0.0], [0.2, 0.3, 2.0]])      # test/gpu_consistency.jl:292


def tri_case(n, dtype, approx=True, seed=42, r_cut=0.8, spread=1.5):
    rng = np.random.default_rng(seed)
    x = (rng.uniform(0, spread, (n, 3))).astype(dtype).astype(np.float64)
    return S.Case(x, np.diag(BASIS), lj=dict(cutoff=("distance", r_cut)), r_list=r_cut, velocities=np.zeros((n, 3)), sigma=np.full(n, 0.3), eps=np.full(n, 1.0),
                  mass=np.full(n, 1.0), triclinic=dict(basis=BASIS, approx_images=approx), name="tri")


def lattice_images(d):
    """all 125 images of displacement d under the lattice, brute force"""
    out = []
    for a in range(-2, 3):
        for b in range(-2, 3):
            for c in range(-2, 3):
                out.append(d + a * BASIS[0] + b * BASIS[1] + c * BASIS[2])
    return np.array(out)


@pytest.mark.parametrize("approx", [True, False])
def test_triclinic_minimum_image_is_the_shortest_lattice_image(approx):
    """for separations under half the smallest box height both variants return THE nearest image; pair forces are equal and opposite"""
    case = tri_case(60, np.float64, approx)
    o = case.oracle(np.float64)
    i, j, sp = o.neighbors("brute")
    assert len(i) > 50
    x = case.coords
    for a, b in list(zip(i, j))[:200]:
        d = x[b] - x[a]
        im = lattice_images(d)
        assert np.sqrt((im ** 2).sum(axis=1).min()) <= 0.8 + 1e-12
    # every pair within the cutoff under brute-force imaging is in the list, and nothing else
    n_ref = 0
    for a in range(case.n):
        for b in range(a):
            if np.sqrt((lattice_images(x[b] - x[a]) ** 2).sum(axis=1).min()) <= 0.8:
                n_ref += 1
    assert n_ref == len(i)
    f = o.forces((i, j, sp))
    assert np.abs(f.sum(axis=0)).max() < 1e-9 * np.abs(f).max()


def test_triclinic_wrap_puts_every_point_inside_the_cell_and_keeps_the_lattice_class():
    """wrap_coords (spatial.jl:588-602): fractional coordinates in [0, 1) afterwards, and the shift is a lattice vector"""
    rng = np.random.default_rng(3)
    case = tri_case(500, np.float64)
    case.coords = rng.uniform(-7, 7, (500, 3))
    o = case.oracle(np.float64)
    o.wrap()
    frac = np.linalg.solve(BASIS.T, o.coords.T).T
    assert frac.min() > -1e-12 and frac.max() < 1 + 1e-12
    shift = np.linalg.solve(BASIS.T, (o.coords - case.coords).T).T
    assert np.abs(shift - np.round(shift)).max() < 1e-9


def test_triclinic_approx_and_exact_images_agree_below_half_the_box_height():
    a, b = tri_case(50, np.float64, True), tri_case(50, np.float64, False)
    oa, ob = a.oracle(np.float64), b.oracle(np.float64)
    na, nb = oa.neighbors("brute"), ob.neighbors("brute")
    assert all(np.array_equal(u, v) for u, v in zip(S.sorted_pairs(*na), S.sorted_pairs(*nb)))
    assert np.abs(oa.forces(na) - ob.forces(nb)).max() < 1e-9
    assert oa.potential_energy(na) == pytest.approx(ob.potential_energy(nb), rel=1e-12)


def test_triclinic_wrap_known_answers_of_the_reference():
    """test/basic.jl:158-190: an orthogonal TriclinicBoundary wraps like the CubicBoundary; for the skewed cell H the result is
    H·(s − floor(s)) with s = H⁻¹·x"""
    lengths = np.array([5.2, 5.1, 5.8])
    pts = np.array([[1.0, 2.0, 3.0], [6.1, -0.2, 12.0], [-5.3, 10.4, -0.1]])
    ortho = S.Case(pts, lengths, lj=dict(cutoff=("distance", 1.0)), r_list=1.0, sigma=np.full(3, 0.3), eps=np.ones(3), mass=np.ones(3),
                   triclinic=dict(basis=np.diag(lengths)))
    o = ortho.oracle(np.float64); o.wrap()
    cubic = S.Case(pts, lengths, lj=dict(cutoff=("distance", 1.0)), r_list=1.0, sigma=np.full(3, 0.3), eps=np.ones(3), mass=np.ones(3))
    oc = cubic.oracle(np.float64); oc.wrap()
    assert np.allclose(o.coords, oc.coords, rtol=1e-12, atol=1e-12)
    H = np.array([[4.0, 0.8, 0.4], [0.0, 3.5, 0.6], [0.0, 0.0, 3.0]])          # columns = basis vectors
    pts = np.array([[1.0, 2.0, 3.0], [5.5, -1.0, 7.2], [-3.0, 8.0, -2.0]])
    skew = S.Case(pts, np.diag(H), lj=dict(cutoff=("distance", 1.0)), r_list=1.0, sigma=np.full(3, 0.3), eps=np.ones(3), mass=np.ones(3),
                  triclinic=dict(basis=H.T))
    o = skew.oracle(np.float64); o.wrap()
    frac = np.linalg.solve(H, pts.T)
    expected = (H @ (frac - np.floor(frac))).T
    assert np.allclose(o.coords, expected, rtol=1e-12, atol=1e-12)


def test_triclinic_free_flight_keeps_velocities_and_stays_wrapped():
    """test/basic.jl:233-260: no interactions, VelocityVerlet without CM removal, 1000 steps in the cell built from lengths
    (2.2, 2.0, 1.8) and angles (50°, 40°, 60°) (:130-135): velocities unchanged, coordinates stay wrapped, no atom jumps"""
    basis = np.array([[2.2, 0.0, 0.0], [1.0, 1.7320508075688772, 0.0], [1.3788800, 0.5399122, 1.0233204]])
    rng = np.random.default_rng(5)
    n = 300
    frac = rng.uniform(0, 1, (n, 3))
    x = frac @ basis
    v = rng.normal(size=(n, 3)) * np.sqrt(8.314462618e-3 * 100.0 / 1.0)
    case = S.Case(x, np.diag(basis), lj=dict(cutoff=("distance", 0.4)), r_list=0.45, velocities=v, sigma=np.full(n, 0.3), eps=np.zeros(n), mass=np.ones(n),
                  triclinic=dict(basis=basis))
    o = case.oracle(np.float64)
    prev = o.coords.copy()
    for k in range(10):
        o.vv_run(100, 0.002, first_step=100 * k, remove_cm_every=0)
        w = o.coords.copy()
        o.wrap()
        assert np.array_equal(o.coords, w)                                   # wrap_coords.(coords) == coords
        d = np.linalg.solve(basis.T, (w - prev - 0.2 * v).T).T                # displacement − v·t is a lattice vector
        assert np.abs(d - np.round(d)).max() < 1e-9
        prev = w
    assert np.allclose(o.vel, v, rtol=1e-12, atol=0)

"""Pins the CPU oracle (oracle/oracle.cpp) to the reference's own known-answer tests.

Every expected number below is copied from the reference's test suite (Molly.jl v0.23.3), cited per test.
"""
import math

import numpy as np
import pytest

from oracle import pyoracle as orc

LJ = dict(lj_enabled=1, lj_cutoff_kind=orc.CUTOFFS["none"])
COUL = dict(coul_kind=orc.COULS["plain"], coul_cutoff_kind=orc.CUTOFFS["none"])


def test_vector_1d_known_answers():
    # test/basic.jl:2-5
    assert orc.vector_1d(4.0, 6.0, 10.0) == 2.0
    assert orc.vector_1d(1.0, 9.0, 10.0) == -2.0
    assert orc.vector_1d(6.0, 4.0, 10.0) == -2.0
    assert orc.vector_1d(9.0, 1.0, 10.0) == 2.0
    for prec in (32, 64):
        # test/basic.jl:11-20 (component-wise, cubic and orthorhombic boxes, exact ==)
        got = [orc.vector_1d(a, b, L, prec) for a, b, L in ((4.0, 6.0, 10.0), (1.0, 9.0, 10.0), (6.0, 4.0, 10.0))]
        assert got == [2.0, -2.0, -2.0]
        got = [orc.vector_1d(a, b, L, prec) for a, b, L in ((4.0, 6.0, 10.0), (1.0, 4.0, 5.0), (1.0, 3.0, 3.5))]
        assert got == [2.0, -2.0, -1.5]


def test_vector_1d_half_box_tie_is_compare_select_form():
    # spatial.jl:491-500: at |dx| == L/2 the select form flips the sign (unlike dx - L*round(dx/L))
    assert orc.vector_1d(1.0, 2.0, 2.0) == -1.0
    assert orc.vector_1d(2.0, 1.0, 2.0) == 1.0


def test_wrap_known_answers():
    # test/basic.jl:32-34
    assert orc.wrap_1d(8.0, 10.0) == 8.0
    assert orc.wrap_1d(12.0, 10.0) == 2.0
    assert orc.wrap_1d(-2.0, 10.0) == 8.0
    assert orc.wrap_1d(3.0, math.inf) == 3.0  # spatial.jl:574-575


def test_mixing_rules():
    # test/interactions.jl:15,18 — Lorentz σ(0.3,0.2)=0.25, geometric ϵ(0.2,0.1)=0.14142135623730953.
    # Checked through the pair energy: V(r=σ_mix) = 0 and V at the minimum 2^(1/6)σ is -ϵ_mix.
    f, pe = orc.pair(LJ, [0.25, 0, 0], si=0.3, sj=0.2, ei=0.2, ej=0.1)
    assert abs(pe) < 1e-15
    rmin = 2 ** (1 / 6) * 0.25
    f, pe = orc.pair(LJ, [rmin, 0, 0], si=0.3, sj=0.2, ei=0.2, ej=0.1)
    assert pe == pytest.approx(-0.14142135623730953, abs=1e-15)
    assert abs(f[0]) < 1e-12


def test_lj_zero_shortcut():
    # mixing.jl:7-11 LJZeroShortcut (default shortcut, lennard_jones.jl:31)
    for kw in (dict(si=0.0, sj=0.3, ei=0.2, ej=0.2), dict(si=0.3, sj=0.3, ei=0.0, ej=0.2)):
        f, pe = orc.pair(LJ, [0.3, 0, 0], **kw)
        assert np.all(f == 0) and pe == 0


@pytest.mark.parametrize("prec,atol", [(64, 1e-9), (32, 2e-5)])
def test_lennard_jones_known_answers(prec, atol):
    # test/interactions.jl:61-82 — a1: σ=0.3, ϵ=0.2; dr12 = 0.3 nm, dr13 = 0.4 nm along x. Same numbers
    # pin LennardJones14 (test/interactions.jl:119-139).
    kw = dict(si=0.3, sj=0.3, ei=0.2, ej=0.2, prec=prec)
    f, pe = orc.pair(LJ, [0.3, 0, 0], **kw)
    assert f == pytest.approx([16.0, 0, 0], abs=atol * 10 if prec == 32 else atol)
    assert pe == pytest.approx(0.0, abs=atol)
    f, pe = orc.pair(LJ, [0.4, 0, 0], **kw)
    assert f == pytest.approx([-1.375509739, 0, 0], abs=atol)
    assert pe == pytest.approx(-0.1170417309, abs=atol)


def test_coulomb_known_answers():
    # test/interactions.jl:374-395 — q = 1, 1
    f, pe = orc.pair(COUL, [0.3, 0, 0], qi=1.0, qj=1.0)
    assert f == pytest.approx([1543.727311, 0, 0], abs=1e-5)
    assert pe == pytest.approx(463.1181933, abs=1e-5)
    f, pe = orc.pair(COUL, [0.4, 0, 0], qi=1.0, qj=1.0)
    assert f == pytest.approx([868.3466125, 0, 0], abs=1e-5)
    assert pe == pytest.approx(347.338645, abs=1e-5)


CUTOFF_ROWS = [  # test/interactions.jl:1587-1594: r = 0.7, rc = 0.8, ra = 0.6, σ = 0.3, ϵ = 0.2
    ("none", -0.04196301990, -0.00492640193),
    ("distance", -0.04196301990, -0.00492640193),
    ("shifted_potential", -0.04196301990, -0.00270785727),
    ("shifted_force", -0.02537033587, -0.00104858887),
    ("cubic_spline", -0.06201171875, -0.00312500000),
    ("polynomial", -0.06716652806, -0.00246320097),
]


@pytest.mark.parametrize("kind,f_ref,pe_ref", CUTOFF_ROWS)
def test_cutoff_known_answers(kind, f_ref, pe_ref):
    inter = dict(lj_enabled=1, lj_cutoff_kind=orc.CUTOFFS[kind], lj_rc=0.8, lj_ra=0.6)
    kw = dict(si=0.3, sj=0.3, ei=0.2, ej=0.2)
    dr12 = [orc.vector_1d(1.0, 1.7, 2.0), 0.0, 0.0]
    f, pe = orc.pair(inter, dr12, **kw)
    assert f[0] == pytest.approx(f_ref, abs=1e-9)
    assert pe == pytest.approx(pe_ref, abs=1e-9)
    if kind != "none":  # :1609-1630 zero beyond the cutoff (r = 1.0 and 0.95)
        for x3 in (2.0, 1.95):
            dr = [orc.vector_1d(1.0, x3, 2.0), 0.0, 0.0]
            f, pe = orc.pair(inter, dr, **kw)
            assert abs(f[0]) < 1e-12 and abs(pe) < 1e-12


def test_ewald_alpha_and_erfc_polynomial():
    # coulomb.jl:1332 α = sqrt(-log(2 tol))/rc = 2.62826... for rc = 1, tol = 5e-4; A&S 7.1.26 max error 1.5e-7
    alpha = math.sqrt(-math.log(2 * 5e-4)) / 1.0
    assert alpha == pytest.approx(2.6282608, abs=1e-6)
    ke = orc.COULOMB_CONST
    for approx, tol in ((1, 2e-7), (0, 1e-14)):
        inter = dict(coul_kind=orc.COULS["ewald_direct"], coul_rc=1.0, ewald_alpha=alpha, ewald_approx_erfc=approx)
        for r in (0.15, 0.4, 0.77, 0.99):
            f, pe = orc.pair(inter, [r, 0, 0], qi=1.0, qj=-0.5)
            assert pe / (ke * -0.5 / r) == pytest.approx(math.erfc(alpha * r), abs=tol)
            g = math.erfc(alpha * r) + 2 * alpha * r * math.exp(-(alpha * r) ** 2) / math.sqrt(math.pi)
            assert f[0] / (ke * -0.5 / r ** 2) == pytest.approx(g, abs=2 * tol)
        f, pe = orc.pair(inter, [1.0001, 0, 0], qi=1.0, qj=-0.5)
        assert f[0] == 0 and pe == 0
        # special pairs: plain Coulomb × weight (coulomb.jl:1411-1414)
        inter["coul_weight_special"] = 0.8333333333333334
        f, pe = orc.pair(inter, [0.3, 0, 0], qi=1.0, qj=1.0, special=True)
        assert f[0] == pytest.approx(1543.727311 * 0.8333333333333334, abs=1e-5)


def test_reaction_field_formula():
    # coulomb.jl:748-814
    rc, eps = 1.0, 78.3
    krf = (1 / rc ** 3) * (eps - 1) / (2 * eps + 1)
    crf = (1 / rc) * (3 * eps) / (2 * eps + 1)
    inter = dict(coul_kind=orc.COULS["reaction_field"], coul_rc=rc, rf_dielectric=eps, coul_weight_special=0.5)
    r, ke = 0.6, orc.COULOMB_CONST
    f, pe = orc.pair(inter, [r, 0, 0], qi=0.4, qj=-0.8)
    assert f[0] == pytest.approx(ke * 0.4 * -0.8 * (1 / r - 2 * krf * r * r) / (r * r) * r, rel=1e-14)
    assert pe == pytest.approx(ke * 0.4 * -0.8 * (1 / r + krf * r * r - crf), rel=1e-14)
    f, pe = orc.pair(inter, [r, 0, 0], qi=0.4, qj=-0.8, special=True)   # krf = crf = 0 and × weight
    assert f[0] == pytest.approx(0.5 * ke * 0.4 * -0.8 / (r * r), rel=1e-14)
    assert pe == pytest.approx(0.5 * ke * 0.4 * -0.8 / r, rel=1e-14)
    inter["rf_dielectric"] = math.inf   # conducting boundary: krf = 1/(2rc³), crf = 3/(2rc)
    f, pe = orc.pair(inter, [r, 0, 0], qi=0.4, qj=-0.8)
    assert pe == pytest.approx(ke * 0.4 * -0.8 * (1 / r + r * r / 2 - 1.5), rel=1e-14)


@pytest.mark.parametrize("method", ["brute", "cell"])
def test_neighbor_list_three_atom_toy(method):
    # test/basic.jl:494-518 → [(1, 2, false)] (1-based) = {(0, 1)} here
    s = orc.OracleSystem([[1, 1, 1], [2, 2, 2], [5, 5, 5]], [10, 10, 10], {}, r_list=2.0)
    i, j, sp = s.neighbors(method)
    assert sorted(zip(np.minimum(i, j), np.maximum(i, j), sp)) == [(0, 1, 0)]


def test_neighbor_cell_equals_brute_random_with_exceptions():
    # test/basic.jl:544-577 pattern: cell finder vs brute force, exact list equality after sorting
    rng = np.random.default_rng(7)
    n, box = 800, np.array([4.0, 5.0, 3.7])
    x = rng.random((n, 3)) * box
    pairs = rng.integers(0, n, (300, 2)); pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    for dtype in (np.float64, np.float32):
        s = orc.OracleSystem(x, box, {}, dtype=dtype, r_list=1.2, excluded=pairs[:150], special=pairs[100:])
        a = s.neighbors("brute"); b = s.neighbors("cell", nthreads=3)
        ka = sorted(zip(a[0].tolist(), a[1].tolist(), a[2].tolist()))
        kb = sorted(zip(b[0].tolist(), b[1].tolist(), b[2].tolist()))
        assert ka == kb and len(ka) > 1000
        assert all(i > j for i, j, _ in ka)   # neighbors.jl:404-408 ordering (i, j<i)


def test_forces_newton_third_law_and_threaded_equals_serial():
    rng = np.random.default_rng(3)
    n, box = 500, np.array([3.0, 3.0, 3.0])
    x = rng.random((n, 3)) * box
    inter = dict(lj_enabled=1, lj_cutoff_kind=1, lj_rc=1.0, coul_kind=2, coul_rc=1.0, rf_dielectric=78.3)
    s = orc.OracleSystem(x, box, inter, charge=rng.normal(size=n) * 0.3, sigma=np.full(n, 0.2), eps=np.full(n, 0.5), r_list=1.2)
    nl = s.neighbors("cell")
    f1 = s.forces(nl)
    f4 = s.forces(nl, nthreads=4)
    assert np.abs(f1.sum(axis=0)).max() < 1e-7 * np.abs(f1).max()
    np.testing.assert_allclose(f4, f1, rtol=1e-10, atol=1e-8)
    # energy is consistent with a central finite difference of the force on one atom
    k, h = 17, 1e-6
    e = []
    for sgn in (+1, -1):
        s.coords[k, 0] += sgn * h
        e.append(s.potential_energy(s.neighbors("cell")))
        s.coords[k, 0] -= sgn * h
    assert -(e[0] - e[1]) / (2 * h) == pytest.approx(f1[k, 0], rel=1e-4, abs=1e-4)


def test_velocity_verlet_chunked_continuation_is_exact():
    # test/simulation.jl:16-57: 10 steps == 3 + 3 + 4 steps with init_step = 3, 6 (exact ==)
    rng = np.random.default_rng(11)
    n, box = 150, np.array([2.5, 2.5, 2.5])
    g = np.stack(np.meshgrid(*[np.arange(6)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n] * 0.4 + 0.1
    x = g + rng.normal(size=(n, 3)) * 0.01
    v = rng.normal(size=(n, 3)) * 0.3
    inter = dict(lj_enabled=1, lj_cutoff_kind=1, lj_rc=1.0)
    kw = dict(sigma=np.full(n, 0.3), eps=np.full(n, 0.2), mass=np.full(n, 10.0), r_list=1.2, rebuild_every=10)
    a = orc.OracleSystem(x, box, inter, velocities=v, **kw)
    a.vv_run(10, 0.002)
    b = orc.OracleSystem(x, box, inter, velocities=v, **kw)
    b.vv_run(3, 0.002, first_step=0); b.vv_run(3, 0.002, first_step=3); b.vv_run(4, 0.002, first_step=6)
    assert np.array_equal(a.coords, b.coords) and np.array_equal(a.vel, b.vel)
    # momentum stays removed (remove_CM_motion=1, simulators.jl:293)
    assert np.abs((a.vel * 10.0).sum(axis=0)).max() < 1e-10


def test_pairwise_virial_obeys_the_scaling_identity():
    """force.jl:848-852: vir += dr ⊗ f.  For a smooth pair potential tr W = Σ dr·f = −dE/dλ under x → λx, L → λL, r_c → λ r_c... here
    with the cutoff fixed and a ShiftedForceCutoff (force and energy continuous at r_c), so the identity holds to O(h²)."""
    from tests import systems as S
    case = S.lj_fluid(6, dtype=np.float64)
    case.lj = dict(cutoff=("shifted_force", 1.0))
    o = case.oracle(np.float64)
    w = o.virial(o.neighbors("cell"))
    assert np.abs(w - w.T).max() < 1e-9 * np.abs(w).max()

    def energy(lam):
        c2 = S.Case(case.coords * lam, case.box * lam, lj=case.lj, r_list=case.r_list, sigma=case.sigma, eps=case.eps, mass=case.mass)
        o2 = c2.oracle(np.float64)
        return o2.potential_energy(o2.neighbors("cell"))

    h = 1e-6
    assert np.trace(w) == pytest.approx(-(energy(1 + h) - energy(1 - h)) / (2 * h), rel=1e-7)


def test_lj_zero_shortcut_reads_lambda():
    """shortcut_pair(::LJZeroShortcut, …) (src/mixing.jl:7-11): a pair is skipped when ϵ, σ OR λ of either atom is zero — the only place the unsoftened
    Lennard-Jones reads Atom.λ (types.jl:466-475).  The oracle with per-atom λ: a λ = 0 atom feels and exerts no LJ force, which is exactly the system
    with that atom's ϵ set to 0; λ = 0.5 changes nothing; the neighbour list does not know λ."""
    from tests import systems as S
    base = S.lj_fluid(6, dtype=np.float64)
    lam = np.ones(base.n); lam[::5] = 0.0; lam[1::7] = 0.5
    mk = lambda **kw: S.Case(base.coords, base.box, lj=base.lj, r_list=base.r_list, sigma=base.sigma, mass=base.mass, **kw).oracle(np.float64)
    eps = np.broadcast_to(np.asarray(base.eps, dtype=np.float64), (base.n,))
    o_lam, o_eps, o_one = mk(eps=eps, lam=lam), mk(eps=np.where(lam == 0, 0.0, eps)), mk(eps=eps)
    nl = o_one.neighbors("cell")
    assert [len(a) for a in o_lam.neighbors("cell")] == [len(a) for a in nl]
    f_lam, f_eps, f_one = o_lam.forces(nl), o_eps.forces(nl), o_one.forces(nl)
    off = lam == 0
    assert np.array_equal(f_lam, f_eps) and np.abs(f_lam[off]).max() == 0.0
    assert np.abs(f_one[off]).max() > 0 and o_lam.potential_energy(nl) == o_eps.potential_energy(nl) != o_one.potential_energy(nl)


def test_memory_limit_recipe_inputs():
    """The system of the reference's "Testing GPU memory limits" example (docs/src/examples.md:969-1000) as molly.jl_amd/workloads.py builds it for
    `bench.py --workload memlimit` and tests/test_gpu_large.py: V = n · 0.013 nm³ and its cube root in Float32, the list radius = the cutoff = 1.0 nm with the
    GPUNeighborFinder's default cadence of 25 steps (src/neighbors.jl:327), zero velocities — and, for uniformly random points in a periodic box, a pair count on the
    closed form N(N−1)/2 · (4/3)π r³ / V that the benchmark holds every size to."""
    import importlib
    from tests import systems as S
    W = importlib.import_module("molly_jl_amd.workloads")
    assert W.memlimit_box(140000) == float(np.cbrt(np.float32(140000) * np.float32(0.013), dtype=np.float32))
    n = 30000
    case = W.memlimit_fluid(n, seed=3)
    assert case.r_list == 1.0 and case.lj["cutoff"] == ("distance", 1.0) and case.rebuild_every == 25 and not case.velocities.any()
    assert np.all(case.sigma == 0.001) and np.all(case.eps == 0.1) and np.all(case.mass == 10.0)
    assert case.coords.min() >= 0 and case.coords.max() < case.box[0]
    oi, oj, _ = case.oracle(np.float32).neighbors("cell", nthreads=4)
    expect = 0.5 * n * (n - 1) * (4.0 / 3.0) * math.pi / float(case.box[0]) ** 3
    assert abs(len(oi) - expect) < 5 * math.sqrt(expect)
    bi, bj, _ = case.oracle(np.float32).neighbors("brute")
    assert np.array_equal(S.pair_keys(oi, oj), S.pair_keys(bi, bj))

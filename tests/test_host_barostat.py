"""Host logic of the boundary-changing couplings (molly.jl_amd/api.py: scale_coords, scale_boundary, MonteCarloBarostat) without a GPU: the three coupling types of
apply_coupling_mc! (coupling.jl:886-1033) restated here line by line and replayed with the same uniform numbers over the ORACLE's potential energies (the product's
barostat takes its energy function as an argument for exactly this; on a GPU it is the engine's, tests/test_gpu_boundary.py)."""
import importlib
import math

import numpy as np
import pytest

from tests import systems as S


def reference_attempt(kind, energy, x, box, rand, volume_scale, P, kT, n):
    """one pass of the `for attempt_n` loop: isotropic coupling.jl:893-932, semiisotropic :943-992, anisotropic :1003-1051 (fp64, no topology)"""
    E = energy(x, box)
    V = float(np.prod(box))
    dV = volume_scale * (2 * rand() - 1)
    v_scale = (V + dV) / V
    if kind == "isotropic":
        l = np.cbrt(v_scale); mu = np.array([l, l, l])
        work = (P[0] + P[1] + P[2]) * dV / 3
    elif kind == "semiisotropic":
        w1, w2 = rand(), rand(); s = w1 + w2; w1, w2 = w1 / s, w2 / s
        mu = np.array([v_scale ** w1, v_scale ** w1, v_scale ** w2])
        work = ((w1 / 2) * P[0] + (w1 / 2) * P[1] + w2 * P[2]) * (V + dV) * math.log(v_scale)
    else:
        w1, w2, w3 = rand(), rand(), rand(); s = w1 + w2 + w3; w1, w2, w3 = w1 / s, w2 / s, w3 / s
        mu = np.array([v_scale ** w1, v_scale ** w2, v_scale ** w3])
        work = (w1 * P[0] + w2 * P[1] + w3 * P[2]) * (V + dV) * math.log(v_scale)
    dW = (energy(x * mu, box * mu) - E) + work - n * kT * math.log(v_scale)
    if dW <= 0 or rand() < math.exp(-dW / kT):
        return True, x * mu, box * mu
    return False, x, box


@pytest.mark.parametrize("kind,pressure", [("isotropic", 1.0), ("semiisotropic", (1.0, 1.0, 40.0)), ("anisotropic", (5.0, 1.0, 60.0))])
def test_monte_carlo_barostat_host_logic_over_the_oracle(pkg, kind, pressure):
    case = S.lj_fluid(7, dtype=np.float64, r_cut=0.9, r_list=1.0)          # 343 atoms, box 2.53 nm
    s = case.system(pkg, np.float64)
    o = case.oracle(np.float64)

    def e_of(x, box):
        o.coords[:] = x
        o.set_boundary(box)
        return o.potential_energy(o.neighbors("cell"))
    api = importlib.import_module("molly_jl_amd.api")
    baro = pkg.MonteCarloBarostat(pressure, 85.0, s.boundary, coupling_type=kind, n_steps=2, scale_factor=0.003)
    vs = baro.volume_scale
    assert vs == pytest.approx(0.003 * float(np.prod(case.box)), rel=1e-15)
    P = np.broadcast_to(np.asarray(pressure, dtype=np.float64), (3,)) * pkg.BAR
    rng_a, rng_b = np.random.default_rng(11), np.random.default_rng(11)
    x, box = case.coords.copy(), case.box.copy()
    got, want, n_att, n_acc = [], [], 0, 0
    for step in range(1, 27):
        r = api._apply_mc_barostat(s, baro, step, rng_a, energy=lambda q: e_of(q.coords, q.boundary.side_lengths))
        if step % 2:
            assert r is False                                              # coupling.jl:864-866: not a barostat step
            continue
        acc, x, box = reference_attempt(kind, e_of, x, box, rng_b.random, vs, P, pkg.BOLTZMANN * 85.0, case.n)
        got.append(r); want.append(acc)
        n_att += 1; n_acc += int(acc)
        if n_att >= 10:
            if n_acc < 0.25 * n_att:
                vs /= 1.1
            elif n_acc > 0.75 * n_att:
                vs = min(vs * 1.1, float(np.prod(box)) * 0.3)
            n_att = n_acc = 0
        assert np.allclose(s.boundary.side_lengths, box, rtol=1e-13, atol=0) and np.abs(s.coords - x).max() < 1e-12, step
    assert got == want and True in got and False in got, (kind, got)
    assert baro.volume_scale == pytest.approx(vs, rel=1e-14) and (baro.n_attempted, baro.n_accepted) == (n_att, n_acc)
    assert s._ctx is None                                                  # no engine was made: host logic only


def test_scale_boundary_and_scale_coords(pkg):
    b = pkg.CubicBoundary(2.0, 3.0, 4.0)
    assert np.array_equal(pkg.scale_boundary(b, 1.5).side_lengths, [3.0, 4.5, 6.0]) and pkg.volume(b) == 24.0      # spatial.jl:414-416, 365
    t = pkg.TriclinicBoundary((2.0, 0, 0), (0.5, 2.0, 0), (0.3, 0.4, 2.0))
    t2 = pkg.scale_boundary(t, (1.0, 2.0, 0.5))
    assert np.allclose(t2.basis_vectors, t.basis_vectors * np.array([1.0, 2.0, 0.5])) and pkg.volume(t) == 8.0      # spatial.jl:420-422
    s = pkg.System(coords=np.array([[0.5, 1.0, 1.5], [1.0, 1.0, 1.0]]), boundary=t, velocities=np.ones((2, 3)), dtype=np.float64)
    pkg.scale_coords(s, np.diag([1.1, 1.1, 1.1]), scale_velocities=True)                                           # spatial.jl:1198-1209
    assert np.allclose(s.coords, [[0.55, 1.1, 1.65], [1.1, 1.1, 1.1]]) and np.allclose(s.velocities, 1 / 1.1)
    assert np.allclose(s.boundary.basis_vectors, t.basis_vectors * 1.1) and s.boundary.approx_images == t.approx_images
    with pytest.raises(ValueError):
        pkg.MonteCarloBarostat((1.0, 1.0, 1.0), 300.0, b)                                                          # coupling.jl:793-795
    with pytest.raises(ValueError):
        pkg.MonteCarloBarostat(1.0, 300.0, b, coupling_type="shear")                                               # :788-790

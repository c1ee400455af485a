"""The oracle's counter-based noise (oracle/stochastic.h): Philox4x32-10 against the Random123 known-answer vectors, the normal
transform's statistics, and the Langevin / Andersen steps through their closed-form properties (CPU only)."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests import systems as S

KB = 8.314462618e-3

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output
PHILOX_KAT = [
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,out", PHILOX_KAT)
def test_philox4x32_10_known_answers(ctr, key, out):
    assert [int(x) for x in orc.philox4x32_10(ctr, key)] == out


def ideal_gas(n, dtype, seed=3, mass=None):
    rng = np.random.default_rng(seed)
    box = 6.0
    x = rng.uniform(0, box, (n, 3)).astype(dtype).astype(np.float64)
    m = np.full(n, 12.0) if mass is None else mass
    return S.Case(x, box, lj=dict(cutoff=("distance", 1.0)), r_list=1.2, velocities=np.zeros((n, 3)), sigma=np.full(n, 0.3), eps=np.zeros(n), mass=m, name="ideal")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_random_velocities_are_maxwell_boltzmann(dtype):
    """random_velocities_kernel! (kernels.jl:688-704): v_i = sqrt(kT/m_i) ξ_i — unit variance, no skew, normal kurtosis, independent
    components; massless atoms stay at rest; a different ctr1 gives a different stream"""
    n = 60000
    mass = np.where(np.arange(n) % 2 == 0, 1.0, 16.0); mass[7] = 0.0
    case = ideal_gas(n, dtype, mass=mass)
    o = case.oracle(dtype)
    o.random_velocities(KB * 300.0, key=0x1234, ctr1=77)
    v = o.vel.astype(np.float64)
    assert np.all(v[7] == 0)
    ok = mass > 0
    z = v[ok] / np.sqrt(KB * 300.0 / mass[ok])[:, None]
    m = ok.sum() * 3
    assert abs(z.mean()) < 4 / np.sqrt(m) and abs(z.var() - 1) < 4 * np.sqrt(2 / m)
    assert abs((z ** 3).mean()) < 4 * np.sqrt(15 / m) and abs((z ** 4).mean() - 3) < 4 * np.sqrt(96 / m)
    c = np.corrcoef(z.T)
    assert np.abs(c - np.eye(3)).max() < 4 / np.sqrt(ok.sum())
    assert np.abs(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]) < 4 / np.sqrt(ok.sum())     # neighbouring counters
    o2 = case.oracle(dtype); o2.random_velocities(KB * 300.0, key=0x1234, ctr1=78)
    assert np.abs(np.corrcoef(o2.vel[ok, 0], v[ok, 0])[0, 1]) < 4 / np.sqrt(ok.sum())
    o3 = case.oracle(dtype); o3.random_velocities(KB * 300.0, key=0x1234, ctr1=77)
    assert np.array_equal(o3.vel, o.vel)                                              # counter based: reproducible


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_andersen_redraws_the_expected_fraction(dtype):
    """apply_andersen_coupling_kernel! (kernels.jl:706-723): rand_u64 < round(prob·2⁶⁴) picks the atom; untouched atoms keep their bits"""
    n = 40000
    case = ideal_gas(n, dtype)
    case.velocities = np.random.default_rng(1).normal(size=(n, 3)).astype(dtype).astype(np.float64)
    for prob in (0.0, 0.25, 1.0):
        o = case.oracle(dtype)
        v0 = o.vel.copy()
        o.andersen(KB * 300.0, prob, key=5, ctr1=9)
        changed = np.any(o.vel != v0, axis=1)
        assert abs(changed.mean() - prob) <= 4 * np.sqrt(prob * (1 - prob) / n)
        if prob == 1.0:    # the velocity comes from the block at ctr0 + natoms: not the stream random_velocities uses
            o2 = case.oracle(dtype); o2.random_velocities(KB * 300.0, key=5, ctr1=9)
            assert not np.array_equal(o2.vel, o.vel)
            z = o.vel / np.sqrt(KB * 300.0 / 12.0)
            assert abs(z.var() - 1) < 4 * np.sqrt(2 / (3 * n))


def test_langevin_without_friction_is_leapfrog_and_with_friction_thermalises():
    """simulators.jl:1171-1197.  friction = 0: vel_scale = 1, noise_scale = 0 → v += a dt; x += v dt, i.e. velocity Verlet's positions with
    half-step-behind velocities.  friction > 0 on an ideal gas: the O-step is an exact Ornstein-Uhlenbeck update, so after t ≫ 1/γ the
    velocities are Maxwell-Boltzmann at the target temperature whatever they started from."""
    case = S.lj_fluid(6, dtype=np.float64)
    dt, n = 0.002, 25
    a = case.oracle(np.float64); a.langevin_run(n, dt, KB * 85.0, 0.0, key=1, ctr1=2, remove_cm_every=0)
    b = case.oracle(np.float64)
    f0 = b.forces(b.neighbors("cell"))
    # velocity Verlet started from v_vv(0) = v(−dt/2) + a(0) dt/2 visits the same positions as the leapfrog started from v(−dt/2)
    c = case.oracle(np.float64)
    c.vel[:] = c.vel + 0.5 * dt * f0 / case.mass[:, None]
    c.vv_run(n, dt, remove_cm_every=0)
    d = a.coords - c.coords; d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 1e-11

    gas = ideal_gas(30000, np.float64)
    gas.velocities = np.full((30000, 3), 0.7)                             # far from equilibrium, with a net drift
    o = gas.oracle(np.float64)
    o.langevin_run(60, 0.01, KB * 250.0, 20.0, key=11, ctr1=0, remove_cm_every=0)   # 12 relaxation times
    z = o.vel / np.sqrt(KB * 250.0 / 12.0)
    m = z.size
    assert abs(z.mean()) < 4 / np.sqrt(m) and abs(z.var() - 1) < 4 * np.sqrt(2 / m)


def test_andersen_coupling_in_velocity_verlet_drives_the_temperature():
    """AndersenThermostat as VelocityVerlet's coupling (simulators.jl:630, coupling.jl:196-211) on the LJ fluid: from 85 K to 300 K"""
    case = S.lj_fluid(8, dtype=np.float64)
    o = case.oracle(np.float64)
    o.set_andersen(KB * 300.0, 0.002 / 0.02, seed=42)
    o.vv_run(60, 0.002, remove_cm_every=1, nthreads=4)
    ke = 0.5 * (case.mass[:, None] * o.vel ** 2).sum()
    t = 2 * ke / ((3 * case.n - 3) * KB)
    assert 240.0 < t < 330.0

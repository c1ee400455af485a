"""test/energy_conservation.jl on the MI355X: 2000 Lennard-Jones atoms (σ = 0.05 nm, ϵ = 0.2 kJ/mol, m = 40) in a 5 nm cubic box at 1 K,
cutoff 3.0 nm (longer than half the box: the exact in-loop minimum image), VelocityVerlet dt = 1 fs without CM removal, the four cutoff
strategies of the reference (:19-24), with and without the neighbour list (:85-86).  Bar as :72: max |E − E0| < 5e-4 kJ/mol over the run, energy
logged every 100 steps (fp64)."""
import numpy as np
import pytest

from tests import systems as S

pytestmark = pytest.mark.gpu


def place_atoms(n, box, min_dist, rng):
    """place_atoms(n, boundary; min_dist) (setup.jl:23-60): uniform rejection sampling under the minimum image"""
    pts = np.empty((0, 3))
    while len(pts) < n:
        c = rng.uniform(0, box, 3)
        d = pts - c
        d -= np.round(d / box) * box
        if len(pts) == 0 or (d ** 2).sum(axis=1).min() > min_dist ** 2:
            pts = np.vstack([pts, c])
    return pts


@pytest.mark.parametrize("use_list", [True, False])
@pytest.mark.parametrize("cutoff", [("distance", 3.0), ("shifted_potential", 3.0), ("shifted_force", 3.0), ("cubic_spline", 3.5, 3.0)])
def test_lennard_jones_energy_conservation(pkg, cutoff, use_list):
    n, box, n_steps = 2000, 5.0, 10000
    rng = np.random.default_rng(17)
    case = S.Case(place_atoms(n, box, 0.1, rng), box, lj=dict(cutoff=cutoff), r_list=3.0 if use_list else np.inf, rebuild_every=10,
                  velocities=np.zeros((n, 3)), sigma=np.full(n, 0.05), eps=np.full(n, 0.2), mass=np.full(n, 40.0), name="lj_dilute")
    if cutoff[0] == "cubic_spline" and use_list:
        case.r_list = 3.5           # the list must reach the spline's outer radius
    s = case.system(pkg, np.float64)
    pkg.random_velocities(s, 1.0, rng=3)
    sim = pkg.VelocityVerlet(dt=0.001, remove_CM_motion=0)
    e0 = pkg.total_energy(s)
    es = []
    for k in range(n_steps // 100):
        pkg.simulate(s, sim, 100, init_step=100 * k)
        es.append(pkg.total_energy(s))
    assert max(abs(e - e0) for e in es) < 5e-4
    assert np.all(s.coords >= 0) and np.all(s.coords < box)

"""Pins the CPU oracle against the reference's OpenMM golden files for 6mrr (data/openmm_6mrr/amber/*, tolerances of
test/protein.jl:263-276: max‖ΔF‖ < 1e-7 kJ/mol/nm, |ΔE| < 1e-5 kJ/mol) and the neighbour count of test/basic.jl:592."""
import numpy as np
import pytest

from tests import golden6mrr as G

FTOL, ETOL = 1e-7, 1e-5


@pytest.fixture(scope="module")
def nl64():
    o = G.case("rf", np.float64, bonded=False).oracle(np.float64)
    return o.neighbors("cell", nthreads=8)


def test_fixture_sanity():
    d = G.data()
    assert d["coords"].shape == (15954, 3)
    assert abs(d["charge"].sum()) < 1e-6 and d["charge"][1] == pytest.approx(0.1642)     # test/protein.jl:141-144
    assert len(d["excluded"]) == 18096
    # masses of the first atoms and the last water (test/protein.jl:186-190): N H H H C … O H H
    assert d["mass"][:5].tolist() == [14.01, 1.008, 1.008, 1.008, 12.01] and d["mass"][-3:].tolist() == [15.99943, 1.007947, 1.007947]
    # KE and temperature of the 300 K velocity file (test/protein.jl:284-286)
    ke = 0.5 * (d["mass"][:, None] * d["velocities_300K"] ** 2).sum()
    assert ke == pytest.approx(65521.87288132431, rel=1.5e-8)   # `≈` in the reference = rtol sqrt(eps)
    assert 2 * ke / ((3 * 15954 - 3) * 8.314462618e-3) == pytest.approx(329.3202932884933, rel=1.5e-8)


def test_neighbor_count_matches_reference(nl64):
    assert len(nl64[0]) == 4602420                                                        # test/basic.jl:592-593
    o32 = G.case("rf", np.float32, bonded=False).oracle(np.float32)
    assert abs(len(o32.neighbors("cell", nthreads=8)[0]) - 4602420) <= 3                  # fp32 inputs: a few boundary pairs may flip


def test_lj_only_forces_and_energy(nl64):
    o = G.case(None, np.float64, bonded=False).oracle(np.float64)
    f = o.forces(nl64, nthreads=8)
    assert np.linalg.norm(f - G.data()["openmm_forces_lj_only"], axis=1).max() < FTOL
    e = o.potential_energy(nl64) + G.lj_dispersion_correction()
    assert abs(e - G.data()["openmm_energy_lj_only"]) < ETOL


def test_coulomb_reaction_field_forces_and_energy(nl64):
    o = G.case("rf", np.float64, bonded=False, lj=False).oracle(np.float64)
    f = o.forces(nl64, nthreads=8)
    assert np.linalg.norm(f - G.data()["openmm_forces_coul_only"], axis=1).max() < FTOL
    assert abs(o.potential_energy(nl64) - G.data()["openmm_energy_coul_only"]) < ETOL * 10   # |E| = 1.2e5: 1e-4 absolute


@pytest.mark.parametrize("term,key", [("bonds", "bond_only"), ("angles", "angle_only"), ("proper", "proptor_only"), ("improper", "improptor_only")])
def test_bonded_terms(term, key):
    o = G.case(None, np.float64, lj=False, which_bonded=(term,)).oracle(np.float64)
    f = o.forces(None, pairwise=False, specific=True)
    assert np.linalg.norm(f - G.data()[f"openmm_forces_{key}"], axis=1).max() < FTOL * 10
    assert abs(o.potential_energy(None, pairwise=False, specific=True) - G.data()[f"openmm_energy_{key}"]) < ETOL


def test_all_cutoff_interactions(nl64):
    o = G.case("rf", np.float64, bonded=True).oracle(np.float64)
    f = o.forces(nl64, nthreads=8, specific=True)
    assert np.linalg.norm(f - G.data()["openmm_forces_all_cut"], axis=1).max() < FTOL * 10
    e = o.potential_energy(nl64, specific=True) + G.lj_dispersion_correction()
    assert abs(e - G.data()["openmm_energy_all_cut"]) < ETOL * 10

"""HIP path on the solvated protein 6mrr (BASELINE.json configs[2] and [4]) against the OpenMM golden files shipped with the
reference (fp64 bars of test/protein.jl:263-276) and against the oracle (fp32, Ewald direct space)."""
import numpy as np
import pytest

from tests import golden6mrr as G
from tests import systems as S

pytestmark = pytest.mark.gpu


def test_neighbor_list_is_the_references(pkg):
    for dtype in (np.float64, np.float32):
        case = G.case("rf", dtype, bonded=False)
        oi, oj, osp = case.oracle(dtype).neighbors("cell", nthreads=8)
        nl = pkg.find_neighbors(case.system(pkg, dtype))
        a, b = S.sorted_pairs(oi, oj, osp), S.sorted_pairs(nl.i, nl.j, nl.special)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        if dtype == np.float64:
            assert nl.n == 4602420                                                       # test/basic.jl:592-593
            assert int(nl.special.sum()) == 3094                                          # 1-4 pairs that are not also 1-2 / 1-3


@pytest.mark.parametrize("key,coul,lj", [("lj_only", None, True), ("coul_only", "rf", False)])
def test_pairwise_terms_vs_openmm_fp64(pkg, key, coul, lj):
    case = G.case(coul, np.float64, bonded=False, lj=lj)
    s = case.system(pkg, np.float64)
    f = pkg.forces(s)
    assert np.linalg.norm(f - G.data()[f"openmm_forces_{key}"], axis=1).max() < 1e-7     # test/protein.jl:267
    e = pkg.potential_energy(s) + (G.lj_dispersion_correction() if lj else 0.0)
    assert abs(e - G.data()[f"openmm_energy_{key}"]) < 1e-4                              # 1e-5 on |E| ~ 5e3; 1e-4 on 1.2e5


@pytest.mark.parametrize("term,key", [("bonds", "bond_only"), ("angles", "angle_only"), ("proper", "proptor_only"), ("improper", "improptor_only")])
def test_bonded_terms_vs_openmm_fp64(pkg, term, key):
    case = G.case(None, np.float64, lj=False, which_bonded=(term,))
    s = case.system(pkg, np.float64)
    f = pkg.forces(s, pairwise=False)
    assert np.linalg.norm(f - G.data()[f"openmm_forces_{key}"], axis=1).max() < 1e-6
    assert abs(pkg.potential_energy(s, pairwise=False) - G.data()[f"openmm_energy_{key}"]) < 1e-5


def test_all_cutoff_interactions_vs_openmm_fp64(pkg):
    case = G.case("rf", np.float64, bonded=True)
    s = case.system(pkg, np.float64)
    f = pkg.forces(s)
    assert np.linalg.norm(f - G.data()["openmm_forces_all_cut"], axis=1).max() < 1e-6
    e = pkg.potential_energy(s) + G.lj_dispersion_correction()
    assert abs(e - G.data()["openmm_energy_all_cut"]) < 1e-4


@pytest.mark.parametrize("coul", ["rf", "ewald"])
def test_pairwise_fp32_vs_fp64_oracle(pkg, coul):
    case = G.case(coul, np.float32, bonded=False)
    tol, o, nl = S.fp32_force_tolerance(case)
    f_ref = o.forces(nl, nthreads=8)
    s = case.system(pkg, np.float32)
    f = pkg.forces(s).astype(np.float64)
    err = np.linalg.norm(f - f_ref, axis=1)
    S.fp32_check(err, tol)
    assert S.rel_rms(err, f_ref) <= max(1.5 * S.fp32_reference_rms(case, f_ref), 5e-6)
    assert pkg.potential_energy(s) == pytest.approx(o.potential_energy(nl), rel=1e-5)


def test_ewald_direct_plus_exclusions_fp64(pkg):
    # CoulombEwald + LJ over the list, bonded terms and the EwaldExclusion list (setup.jl:1877-1913), exact erfc
    case = G.case("ewald", np.float64, bonded=True, approx_erfc=False)
    o = case.oracle(np.float64)
    nl = o.neighbors("cell", nthreads=8)
    f_ref = o.forces(nl, nthreads=8, specific=True)
    s = case.system(pkg, np.float64)
    f = pkg.forces(s)
    assert np.linalg.norm(f - f_ref, axis=1).max() < 1e-6
    assert pkg.potential_energy(s) == pytest.approx(o.potential_energy(nl, specific=True), rel=1e-11)
    # the same in fp32 with the A&S erfc polynomial (the configuration bench.py times)
    case32 = G.case("ewald", np.float32, bonded=True)
    tol, o32, nl32 = S.fp32_force_tolerance(case32)
    f_ref = o32.forces(nl32, nthreads=8, specific=True)
    bonded_scale = np.linalg.norm(o32.forces(None, pairwise=False, specific=True), axis=1)
    f = pkg.forces(case32.system(pkg, np.float32)).astype(np.float64)
    S.fp32_check(np.linalg.norm(f - f_ref, axis=1), tol + 2e-5 * bonded_scale + 2e-3, "fp32 pair + bonded forces (bar + 2e-5·bonded scale + 2e-3)")


def test_velocity_verlet_rf_fp64_tracks_oracle_and_conserves_energy(pkg):
    # BASELINE.json configs[4]: reaction-field Coulomb, Float64, NVE (remove_CM_motion = 0), dt = 0.5 fs
    case = G.case("rf", np.float64, bonded=True)
    o = case.oracle(np.float64)
    o.vv_run(20, 0.0005, remove_cm_every=0, nthreads=8, specific=True)
    s = case.system(pkg, np.float64)
    e0 = pkg.total_energy(s)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.0005, remove_CM_motion=0), 20)
    assert np.abs(s.coords - o.coords).max() < 1e-9 and np.abs(s.velocities - o.vel).max() < 1e-7
    # 0.1 ps of NVE: the unconstrained O-H stretches of the OpenMM-equilibrated start ring coherently (KE swings between
    # 0.6e5 and 2.4e5 kJ/mol with a 50 fs beat), and velocity Verlet at ω·dt ≈ 0.35 shadows that by (ω dt)²/8 ≈ 1.5 % of the
    # vibrational energy — so the bar is boundedness at equal phase, not a flat line (SURVEY §8(d) cfg 5: "report, do not gate").
    es = [e0]
    for k in range(1, 10):
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.0005, remove_CM_motion=0), 20, init_step=20 * k)
        es.append(pkg.total_energy(s))
    assert max(abs(e - e0) for e in es) < 0.06 * abs(e0)            # bounded
    assert abs(es[4] - e0) < 2e-3 * abs(e0) and abs(es[-1] - e0) < 5e-3 * abs(e0)   # returns at equal phase (steps 100, 200): no secular drift


def test_group_split_pair_pass_is_the_same_run(pkg, monkeypatch):
    """The pair pass of this system class runs as four 256-lane workgroups per block over tile quarters, with the charge spreading and the
    bonded terms as further workgroups of the same launch (csrc/forces_gs.hip), the partial forces folded by the bonded sums.  It walks the
    same inner list re-dealt (k_regroup, after every prune) with the same arithmetic per pair, so a run with it and a run with the one-block
    launch (MOLLYHIP_GROUP_SPLIT=0) may differ by the order of an atom's partial sums only: 12 steps of 0.5 fs across a rebuild (search,
    prune, regroup at step 10) agree to fp32 round-off, and the launch is reproducible bit for bit."""
    def run(gs):
        if gs is None: monkeypatch.delenv("MOLLYHIP_GROUP_SPLIT", raising=False)
        else: monkeypatch.setenv("MOLLYHIP_GROUP_SPLIT", gs)
        case = G.case("ewald", np.float32, bonded=True, pme=True)
        s = case.system(pkg, np.float32)
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.0005), 12)
        return np.array(s.coords, dtype=np.float64), np.array(s.velocities, dtype=np.float64), s.stats()
    x1, v1, st1 = run(None)
    x0, v0, st0 = run("0")
    assert st1["group_split"] == 4 and st1["n_group_split_passes"] >= 8 and st0["n_group_split_passes"] == 0
    d = x1 - x0; d -= np.round(d / G.data()["box"]) * G.data()["box"]
    assert np.abs(d).max() < 2e-6 and np.abs(v1 - v0).max() < 2e-3           # (ulp of a 5 nm coordinate: 4.8e-7 nm; hydrogens move at 3 nm/ps)
    # reaction field, no PME: the bonded terms alone ride with the pair groups.  No atomics anywhere on this path (the PME mesh above is
    # flushed with float atomics): the run repeats bit for bit.
    monkeypatch.delenv("MOLLYHIP_GROUP_SPLIT", raising=False)
    case = G.case("rf", np.float32, bonded=True)
    o = case.oracle(np.float64)
    o.vv_run(12, 0.0005, remove_cm_every=1, nthreads=8, specific=True)
    runs = []
    for _ in range(2):
        s = case.system(pkg, np.float32)
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.0005), 12)
        assert s.stats()["n_group_split_passes"] >= 8
        runs.append((np.array(s.coords), np.array(s.velocities)))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    d = runs[0][0].astype(np.float64) - o.coords; d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 5e-6


def test_outer_list_adopted_when_the_prune_has_nothing_to_drop(pkg, monkeypatch):
    """At 0.5 fs the hydrogens outrun a 0.1 nm inner skin within a check interval, the engine raises the skin to the reference's own r_list − cutoff
    (0.2 nm) and drops the outer margin: every rebuild is then a search with r_list followed by a prune that keeps everything.  That prune is skipped —
    the outer list is dealt to the groups directly and read wherever the inner list would be (engine.hip, inner_is_outer).  Same pairs, same
    arithmetic: 120 steps with the adoption and without it (MOLLYHIP_ADOPT_OUTER=0) agree at the fp32 trajectory level."""
    def run(adopt):
        if adopt: monkeypatch.delenv("MOLLYHIP_ADOPT_OUTER", raising=False)
        else: monkeypatch.setenv("MOLLYHIP_ADOPT_OUTER", "0")
        case = G.case("rf", np.float32, bonded=True)
        s = case.system(pkg, np.float32)
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.0005), 120)
        return case, s, s.stats()
    case, s1, st1 = run(True)
    _, s0, st0 = run(False)
    assert st1["n_adopted_outer_lists"] >= 2 and st0["n_adopted_outer_lists"] == 0, (st1, st0)
    assert st1["n_group_split_passes"] > st0["n_group_split_passes"]          # the rebuild steps run as group-split passes too
    d = np.array(s1.coords, dtype=np.float64) - np.array(s0.coords, dtype=np.float64); d -= np.round(d / case.box) * case.box
    assert np.abs(d).max() < 2e-4, np.abs(d).max()                            # 120 steps of fp32 dynamics from different summation orders at the rebuild steps
    assert np.abs(np.array(s1.velocities, dtype=np.float64) - np.array(s0.velocities, dtype=np.float64)).max() < 0.05


def test_calls_behind_a_run_that_adopted_its_outer_list(pkg, monkeypatch):
    """what reads "the inner list" while the outer list stands in for it (engine.hip, inner_is_outer): an energy pass, a force call through the drop-in
    boundary and the exported pair list, on the state a 120-step run ended in — each against the fp64 oracle on the same coordinates"""
    monkeypatch.delenv("MOLLYHIP_ADOPT_OUTER", raising=False)
    case = G.case("rf", np.float32, bonded=True)
    s = case.system(pkg, np.float32)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.0005), 120)
    assert s.stats()["n_adopted_outer_lists"] >= 2
    L = pkg.lib()
    x = np.array(s.coords, dtype=np.float64)
    o = case.oracle(np.float64, coords=x)
    nl = o.neighbors("cell", nthreads=8)
    # straight through the C ABI, no new state: the lists of the run are the lists of these calls
    import ctypes as C
    f = np.zeros((case.n, 3), np.float32)
    s._check(L.mhip_forces(s.engine(), 120, 0, s._ptr(f), None, 0))      # (MEM_HOST)
    f_ref = o.forces(nl, nthreads=8, specific=False)
    err = np.linalg.norm(f.astype(np.float64) - f_ref, axis=1)
    assert err.max() < 2e-4 * np.linalg.norm(f_ref, axis=1).max(), err.max()
    e = C.c_double(0)
    s._check(L.mhip_potential_energy(s.engine(), 120, C.byref(e)))
    e_ref = o.potential_energy(nl)
    assert abs(e.value - e_ref) < 2e-5 * abs(e_ref), (e.value, e_ref)
    keys, n_special = S.export_keys(pkg, s)
    oi, oj, osp = case.oracle(np.float32, coords=x).neighbors("cell", nthreads=8)
    assert np.array_equal(keys, S.pair_keys(oi, oj)) and n_special == int(np.asarray(osp).sum())


@pytest.mark.parametrize("remove_cm", [1, 0])
def test_integrator_inside_the_last_force_launch_is_the_same_run(pkg, monkeypatch, remove_cm):
    """Mid-run steps of the complete PME configuration integrate inside their last force launch (step_fused.h, k_gather_collect_vv: interpolation + bonded sums +
    velocity Verlet, v_cm from the one partial the pair launch's extra workgroup leaves) instead of k_gather_collect + k_vv_mid.  Same sums in the same order, same
    integrator helpers: 40 steps of 0.5 fs across two rebuilds agree with the two-launch form (MOLLYHIP_FUSE_GATHER_VV=0) to fp32 round-off — not bit for bit only
    because the charge mesh is flushed with float atomics, which no two runs of either form repeat."""
    def run(fuse):
        monkeypatch.setenv("MOLLYHIP_FUSE_GATHER_VV", fuse)
        case = G.case("ewald", np.float32, bonded=True, pme=True)
        s = case.system(pkg, np.float32)
        sim = pkg.VelocityVerlet(dt=0.0005, remove_CM_motion=remove_cm)
        pkg.simulate(s, sim, 25)
        pkg.simulate(s, sim, 15, init_step=25)
        return np.array(s.coords, dtype=np.float64), np.array(s.velocities, dtype=np.float64), s.stats()
    x1, v1, st1 = run("1")
    x0, v0, st0 = run("0")
    assert st1["n_fused_steps"] >= 30 and st0["n_fused_steps"] == 0, (st1["n_fused_steps"], st0["n_fused_steps"])
    d = x1 - x0; d -= np.round(d / G.data()["box"]) * G.data()["box"]
    assert np.abs(d).max() < 4e-6 and np.abs(v1 - v0).max() < 4e-3, (np.abs(d).max(), np.abs(v1 - v0).max())

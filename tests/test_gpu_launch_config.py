"""mhip_optimize_launch_config / mhip_set_launch_config ≙ optimize_cuda_launch_config! / set_cuda_launch_config! (src/cuda_config.jl:17-62,
ext/MollyCUDAExt.jl:594-642): the tuned or chosen workgroup shape changes how the work is cut, never the result."""
import numpy as np
import pytest

from tests import systems as S

pytestmark = pytest.mark.gpu


def _forces(pkg, s):
    return pkg.forces(s, step_n=0)


def test_optimize_launch_config_times_candidates_and_installs_the_fastest(pkg):
    case = S.lj_fluid(16, dtype=np.float64)                   # 4096 atoms
    s = case.system(pkg, np.float64)
    f0 = _forces(pkg, s)
    auto = (s.stats()["block_atoms"], s.stats()["j_split"])
    trials = pkg.optimize_launch_config(s, n_passes=5)
    timed = [t for t in trials if t[2] > 0]
    assert len(timed) >= 3 and len({t[:2] for t in trials}) == len(trials)           # several shapes, each once
    assert all(bi in (64, 128, 256) and bi * js <= 512 for bi, js, _ in trials)      # fp64: 512 lanes per workgroup
    best = min(timed, key=lambda t: t[2])
    f1 = _forces(pkg, s)
    st = s.stats()
    assert (st["block_atoms"], st["j_split"]) == best[:2]
    tol = 1e-9 * np.abs(f0).max()
    assert np.abs(f1 - f0).max() < tol                        # another summation order, the same forces
    o = case.oracle(np.float64)
    fo = o.forces(o.neighbors("cell"))
    assert np.abs(f1 - fo).max() < 1e-8 * np.abs(fo).max()
    # an explicit shape, then back to the automatic one
    pkg.set_launch_config(s, 64, 8)
    f2 = _forces(pkg, s)
    st = s.stats()
    assert (st["block_atoms"], st["j_split"]) == (64, 8) and np.abs(f2 - f0).max() < tol
    pkg.set_launch_config(s)
    f3 = _forces(pkg, s)
    st = s.stats()
    assert (st["block_atoms"], st["j_split"]) == auto and np.array_equal(f3, f0)      # same shape, same lists, same bits
    with pytest.raises(pkg.MollyHipError):
        pkg.set_launch_config(s, 100, 2)
    with pytest.raises(pkg.MollyHipError):
        pkg.set_launch_config(s, 256, 4)                      # 1024 lanes: over the fp64 launch bound


def test_tuned_shape_survives_a_run_fp32(pkg):
    case = S.lj_fluid(20, dtype=np.float32)                   # 8000 atoms, the packed one-type loop
    s = case.system(pkg, np.float32)
    trials = pkg.optimize_launch_config(s, n_passes=5)
    best = min((t for t in trials if t[2] > 0), key=lambda t: t[2])
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), 40)
    st = s.stats()
    assert (st["block_atoms"], st["j_split"]) == best[:2]
    o = case.oracle(np.float64)
    o.vv_run(40, 0.002, remove_cm_every=1)
    d = s.coords.astype(np.float64) - o.coords
    d -= np.round(d / case.box) * case.box
    assert np.abs(d).mean() < 5e-4                            # the bar of test/simulation.jl:625 for fp32 runs

"""CPU-side checks: the C-ABI library loads and exports every symbol include/mollyhip.h declares, the host
mirror validates its inputs like the reference, and the product path fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mollyhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    assert os.path.exists(pkg.LIB_PATH), "libmollyhip.so not built: run python -c 'import __graft_entry__ as g; g.build()'"
    lib = pkg.lib()
    declared = header_symbols()
    assert len(declared) >= 30
    assert sorted(pkg.SIGNATURES) == declared
    for name in declared:
        assert getattr(lib, name) is not None


def test_ctypes_signatures_have_the_header_arity(pkg):
    """every binding passes as many arguments as the C declaration takes (a changed prototype must not go unnoticed in the ctypes layer)"""
    text = open(os.path.join(ROOT, "include", "mollyhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = dict(re.findall(r"\b(mhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert set(decls) == set(pkg.SIGNATURES)
    for name, params in decls.items():
        params = params.strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(pkg.SIGNATURES[name][1]), (name, params)


def test_struct_layouts_match_header(pkg):
    import ctypes as C
    # sizes computed from the C declarations (LP64): see include/mollyhip.h
    assert C.sizeof(pkg.Interactions) == 4 * 2 + 8 * 3 + 4 * 2 + 8 * 6 + 4 * 2
    assert C.sizeof(pkg.Config) == 4 * 2 + 8 + 24 + 24 + 12 + 4 + 8 + C.sizeof(pkg.Interactions)
    assert C.sizeof(pkg.Stats) == 8 * 9 + 4 * 4 + 8 * 3 + 8 + 8 * 8 + 8 * 8 + 24 + 8 + 4 * 2 + 8 + 8 * 4 + 8      # (… + n_group_split_passes, group_split, n_adopted_outer_lists, n_fused_steps, the four list-upkeep figures, n_box_changes)
    from molly_jl_amd import _lib
    assert C.sizeof(_lib.HaloPlan) == 8 * 2 + 8 * 2 + 4 * 2 + 8 * 2 + 8 + 8 + 8 + 4 + 4     # mhip_halo_plan (…, n_send_cm + tail padding)


def test_product_path_fails_loudly_without_gpu(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    s = pkg.System(coords=np.random.rand(10, 3), boundary=pkg.CubicBoundary(2.0), pairwise_inters=(pkg.LennardJones(),))
    with pytest.raises(pkg.MollyHipError) as e:
        pkg.forces(s)
    assert e.value.code == -5   # MHIP_ERR_NO_DEVICE: no CPU fallback exists


def test_create_rejects_bad_config(pkg):
    import ctypes as C
    cfg = pkg.Config()
    cfg.precision = 16; cfg.n_atoms = 10
    ctx = C.c_void_p()
    assert pkg.lib().mhip_create(C.byref(ctx), C.byref(cfg)) < 0
    assert pkg.lib().mhip_last_error(None)
    assert pkg.lib().mhip_forces(None, 0, 0, None, None, 0) == -1   # null context


def test_host_mirror_validation(pkg):
    with pytest.raises(ValueError):
        pkg.CubicSplineCutoff(0.8, 0.6)          # cutoffs.jl:180-183
    with pytest.raises(ValueError):
        pkg.CubicBoundary(1.0, -1.0, 1.0)
    with pytest.raises(ValueError):
        pkg.System(atoms=[pkg.Atom()] * 3, coords=np.zeros((4, 3)), boundary=pkg.CubicBoundary(2.0))   # types.jl:914
    with pytest.raises(ValueError):
        pkg.System(coords=np.zeros((4, 3)), velocities=np.zeros((3, 3)), boundary=pkg.CubicBoundary(2.0))
    # α = sqrt(-log(2·tol))/rc (coulomb.jl:1332): 2.6282608 for rc = 1, tol = 5e-4
    assert pkg.CoulombEwald(dist_cutoff=1.0).α == pytest.approx(2.6282608, abs=1e-6)
    # normalize_pairs: i<j, sorted, unique, self-pairs dropped (neighbors.jl:171-195, test/basic.jl:701-737)
    nf = pkg.GPUNeighborFinder(dist_cutoff=1.2, excluded_pairs=[[3, 1], [1, 3], [2, 2], [0, 5]], special_pairs=[[4, 2]])
    assert nf.excluded.tolist() == [[0, 5], [1, 3]] and nf.special.tolist() == [[2, 4]]
    el = np.ones((4, 4), bool); el[0, 1] = el[1, 0] = False; np.fill_diagonal(el, False)
    nf = pkg.GPUNeighborFinder(dist_cutoff=1.2, eligible=el)
    assert nf.excluded.tolist() == [[0, 1]]
    s = pkg.System(coords=np.zeros((4, 3)), boundary=pkg.CubicBoundary(2.0),
                   pairwise_inters=(pkg.LennardJones(cutoff=pkg.DistanceCutoff(1.0), use_neighbors=True, weight_special=0.5),
                                    pkg.CoulombReactionField(dist_cutoff=1.0, use_neighbors=True)))
    it = s.interactions()
    assert (it.lj_enabled, it.lj_cutoff_kind, it.lj_rc, it.lj_weight_special) == (1, 1, 1.0, 0.5)
    assert (it.coul_kind, it.coul_rc, it.rf_dielectric) == (2, 1.0, 78.3)
    with pytest.raises(pkg.MollyHipError):
        pkg.System(coords=np.zeros((4, 3)), boundary=pkg.CubicBoundary(2.0), pairwise_inters=(object(),)).interactions()


def test_host_mirror_of_the_widened_rows(pkg):
    """TriclinicBoundary's constructor checks (test/basic.jl:202-212 → ArgumentError), the host wrap_coords against the oracle's, the
    Langevin constants (simulators.jl:1091-1093) and the simulators / couplings simulate() accepts"""
    from tests import systems as S
    with pytest.raises(ValueError):
        pkg.TriclinicBoundary((2.0, 1.0, 0.0), (1.0, 2.0, 0.0), (1.0, 1.0, 2.0))
    with pytest.raises(ValueError):
        pkg.TriclinicBoundary((2.0, 0.0, 0.0), (1.0, 2.0, 0.5), (1.0, 1.0, 2.0))
    with pytest.raises(ValueError):
        pkg.TriclinicBoundary((2.0, 0.0, 0.0), (1.0, 2.0, 0.0), (1.0, 1.0, -2.0))
    basis = np.array([[2.2, 0.0, 0.0], [1.0, 1.7320508075688772, 0.0], [1.37888, 0.5399122, 1.0233204]])
    b = pkg.TriclinicBoundary(*basis)
    assert b.side_lengths.tolist() == [2.2, 1.7320508075688772, 1.0233204] and b.approx_images
    assert np.prod(b.side_lengths) == pytest.approx(3.89937463181886, rel=1e-6)          # volume(b), test/basic.jl:193
    pts = np.random.default_rng(2).uniform(-9, 9, (400, 3))
    case = S.Case(pts, np.diag(basis), lj=dict(cutoff=("distance", 0.4)), r_list=0.45, sigma=np.full(400, 0.3), eps=np.zeros(400), mass=np.ones(400),
                  triclinic=dict(basis=basis))
    o = case.oracle(np.float64); o.wrap()
    assert np.array_equal(pkg.wrap_coords(pts, b), o.coords)                              # same operations in the same order
    assert np.array_equal(pkg.wrap_coords(pts, pkg.CubicBoundary(2.0)), pts - np.floor(pts / 2.0) * 2.0)

    sim = pkg.Langevin(dt=0.002, temperature=300.0, friction=1.5)
    assert sim.vel_scale == np.exp(-0.002 * 1.5) and sim.noise_scale == np.sqrt(1 - sim.vel_scale ** 2) and sim.remove_CM_motion == 1
    s = pkg.System(coords=np.zeros((4, 3)), boundary=pkg.CubicBoundary(2.0), pairwise_inters=(pkg.LennardJones(),))
    with pytest.raises(pkg.MollyHipError):
        pkg.simulate(s, object(), 1)                                                      # unknown simulator
    with pytest.raises(pkg.MollyHipError):
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.001, coupling=object()), 1)               # only AndersenThermostat couples
    with pytest.raises(ValueError):
        pkg.simulate(s, pkg.VelocityVerlet(dt=0.001), 1, init_step=-1)
    with pytest.raises(pkg.MollyHipError):
        pkg.apply_coupling(s, object(), sim)


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """the drop-in boundary is a C ABI: include/mollyhip.h must compile as C99 (no C++ in the signatures), and a plain C client that
    takes the address of every entry point links against libmollyhip.so"""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "mollyhip.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    names = sorted(set(re.findall(r"\b(mhip_[a-z0-9_]+)\s*\(", open(hdr).read())))
    assert len(names) >= 55
    src = tmp_path / "client.c"
    src.write_text('#include "mollyhip.h"\n#include <stdio.h>\nint main(void) {\n  const void* p[] = {' + ", ".join(f"(const void*)&{n}" for n in names) +
                   '};\n  unsigned i, n = 0; for (i = 0; i < sizeof p / sizeof p[0]; ++i) n += p[i] != 0;\n  printf("%u\\n", n); return 0; }\n')
    exe = tmp_path / "client"
    lib_dir = os.path.join(root, "molly.jl_amd")
    subprocess.run(["gcc", "-std=c99", "-Wno-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", lib_dir, "-l:libmollyhip.so",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,--allow-shlib-undefined"], check=True)

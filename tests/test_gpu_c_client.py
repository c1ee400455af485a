"""The C ABI driven by a C program (tests/c_client/mhip_drive.c, gcc, no Python in the call path): create → set_atoms → set_state →
forces / energies → vv_run → get_state → set_state + forces(step_n) × 5 → stats → set_box (a trial move and back) → destroy.  Its numbers are checked against the CPU
oracle on the same inputs: fp64 forces and energies at the reference's bars (test/protein.jl:267, 274), the 20-step trajectory at
1e-9 nm, and the drop-in cadence (no new neighbour search for unchanged coordinates)."""
import os
import subprocess

import numpy as np
import pytest

from tests import systems as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_drives_the_engine(tmp_path):
    exe, out = tmp_path / "mhip_drive", tmp_path / "out.bin"
    lib_dir = os.path.join(ROOT, "molly.jl_amd")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_client", "mhip_drive.c"),
                    "-o", str(exe), "-L", lib_dir, "-l:libmollyhip.so", f"-Wl,-rpath,{lib_dir}", "-Wl,--allow-shlib-undefined", "-lm"], check=True)
    n_side, n_steps = 12, 20
    r = subprocess.run([str(exe), str(n_side), str(n_steps), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = np.fromfile(out, dtype=np.float64)
    n = int(raw[0]); box, dt = raw[1], raw[2]
    pe0, ke0, pe1, ke1 = raw[4:8]
    body = raw[8:8 + 18 * n].reshape(6, n, 3)
    x0, v0, f0, x1, v1, f1 = body
    searches, prunes, calls, pairs_full, pe_scaled, pe_back = raw[8 + 18 * n:]
    assert n == n_side ** 3 and int(raw[3]) == n_steps
    case = S.Case(x0, box, lj=dict(cutoff=("distance", 1.0)), r_list=1.2, rebuild_every=10, velocities=v0,
                  sigma=np.full(n, 0.34), eps=np.full(n, 0.997), mass=np.full(n, 39.948))
    o = case.oracle(np.float64)
    nl = o.neighbors("cell")
    f_ref = o.forces(nl)
    assert np.abs(f0 - f_ref).max() < 1e-7                                   # kJ/mol/nm, test/protein.jl:267
    assert pe0 == pytest.approx(o.potential_energy(nl), rel=1e-10, abs=1e-6)         # the reference's bar is 1e-5 kJ/mol on ~1e5 (protein.jl:274)
    assert ke0 == pytest.approx(o.kinetic_energy(), rel=1e-12)
    o.vv_run(n_steps, dt, remove_cm_every=1)
    d = x1 - o.coords
    d -= np.round(d / box) * box
    assert np.abs(d).max() < 1e-9 and np.abs(v1 - o.vel).max() < 1e-8
    assert ke1 == pytest.approx(o.kinetic_energy(), rel=1e-9)
    # the set_state → forces(step_n) loop on unchanged coordinates: correct forces, and no search was needed for them
    o2 = case.oracle(np.float64, coords=x1)
    nl2 = o2.neighbors("cell")
    f1_ref = o2.forces(nl2)
    assert int(pairs_full) == 2 * len(nl2[0])                                # the statistics count the reference's list of the final coordinates
    assert np.abs(f1 - f1_ref).max() < 1e-7
    assert int(calls) == 5 and int(searches) == 0 and int(prunes) <= 1
    # mhip_set_box: the energy on the box scaled by 1 % (coordinates with it), and the old energy after the move is taken back
    o3 = S.Case(1.01 * x1, 1.01 * box, lj=dict(cutoff=("distance", 1.0)), r_list=1.2, rebuild_every=10, sigma=np.full(n, 0.34), eps=np.full(n, 0.997),
                mass=np.full(n, 39.948)).oracle(np.float64)
    assert pe_scaled == pytest.approx(o3.potential_energy(o3.neighbors("cell")), rel=1e-10, abs=1e-6)
    assert pe_back == pytest.approx(pe1, rel=1e-11)

"""The solvated protein 6mrr, 15 954 atoms, Amber ff99SB-ILDN + TIP3P (both files built by tools/param_6mrr.py from the reference's data files):
the workload's inputs from molly.jl_amd/data/6mrr_system.npz, the OpenMM Reference-platform forces, energies and 100-step trajectory from
tests/golden/6mrr.npz."""
import os

import numpy as np

from tests import systems as S

_W = S._W


_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "6mrr.npz")
_cache = {}


def data():
    if not _cache:
        _cache.update(_W.protein_6mrr_data())
        _cache.update(dict(np.load(_GOLDEN)))
    return _cache


def case(*args, **kw):
    """the 6mrr workload (molly.jl_amd/workloads.py `protein_6mrr`: same arguments) with the checker's side attached by tests/systems.py"""
    return _W.protein_6mrr(*args, **kw)


def lj_dispersion_correction(d=None, rc=1.0):
    """LJDispersionCorrection energy (lennard_jones.jl:165-246): the reference includes it in the lj_only and all_cut
    energies (test/protein.jl:247-248); forces are unaffected."""
    d = data() if d is None else d
    sig, eps = d["sigma"], d["eps"]
    n = len(sig)
    # Σ_{i>=j} ϵ_ij σ_ij^6 over atom TYPES (Lorentz σ, geometric ϵ)
    types, inv, counts = np.unique(np.stack([sig, eps], 1), axis=0, return_inverse=True, return_counts=True)
    s6 = s12 = 0.0
    for a in range(len(types)):
        for b in range(a + 1):
            s = 0.5 * (types[a, 0] + types[b, 0]); e = np.sqrt(types[a, 1] * types[b, 1])
            npair = counts[a] * counts[b] if a != b else counts[a] * (counts[a] + 1) // 2
            s6 += npair * e * s ** 6; s12 += npair * e * s ** 12
    npairs = n * (n + 1) // 2
    f6 = 8 * np.pi * n ** 2 * (-(s6 / npairs) / (3 * rc ** 3))
    f12 = 8 * np.pi * n ** 2 * ((s12 / npairs) / (9 * rc ** 9))
    return (f6 + f12) / float(np.prod(d["box"]))

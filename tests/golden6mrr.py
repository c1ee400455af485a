"""Loader of tests/golden/6mrr.npz (built by tools/param_6mrr.py from the reference's data files): the solvated
protein 6mrr, 15 954 atoms, Amber ff99SB-ILDN + TIP3P, with the OpenMM Reference-platform forces and energies."""
import os

import numpy as np

from tests import systems as S

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "6mrr.npz")
_cache = {}


def data():
    if "d" not in _cache:
        _cache["d"] = dict(np.load(_PATH))
    return _cache["d"]


def case(coulomb="rf", dtype=np.float64, bonded=True, lj=True, which_bonded=("bonds", "angles", "proper", "improper"), r_list=1.2,
         approx_erfc=True, rebuild_every=10, pme=False):
    """coulomb: None | "rf" (CoulombReactionField rc 1.0, ε 78.3 — OpenMM CutoffPeriodic) | "ewald" (CoulombEwald rc 1.0,
    tol 5e-4 + EwaldExclusion list), as setup.jl:1852-1913 wires them for nonbonded_method :cutoff / :pme."""
    d = data()
    T = np.dtype(dtype).type
    r = lambda a: np.asarray(a, dtype=np.float64).astype(dtype).astype(np.float64)   # inputs rounded to the working precision
    coords = r(d["coords"]); box = r(d["box"])
    coords = np.where(coords >= box, 0.0, coords)
    coul = None
    if coulomb == "rf":
        coul = dict(kind="rf", rc=1.0, eps_rf=78.3, weight_special=float(d["weight_14_coulomb"]))
    elif coulomb == "ewald":
        coul = dict(kind="ewald", rc=1.0, tol=5e-4, approx=approx_erfc, weight_special=float(d["weight_14_coulomb"]))
    kw = {}
    if bonded:
        if "bonds" in which_bonded:
            kw["bonds"] = dict(i=d["bonds_i"], j=d["bonds_j"], k=d["bonds_k"], r0=d["bonds_r0"])
        if "angles" in which_bonded:
            kw["angles"] = dict(i=d["angles_i"], j=d["angles_j"], k=d["angles_k"], kth=d["angles_kth"], th0=d["angles_th0"])
        parts = [p for p in ("proper", "improper") if p in which_bonded]
        if parts:
            kw["torsions"] = {k: np.concatenate([d[f"{p}_{k}"] for p in parts]) for k in ("i", "j", "k", "l", "periodicity", "phase", "k0")}
        if coulomb == "ewald":
            kw["ewald_excl"] = d["ewald_excl"]
    return S.Case(coords, box, lj=dict(cutoff=("distance", 1.0), weight_special=float(d["weight_14_lj"])) if lj else None, coul=coul,
                  r_list=r_list, rebuild_every=rebuild_every, velocities=r(d["velocities_300K"]), charge=r(d["charge"]), sigma=r(d["sigma"]),
                  eps=r(d["eps"]), mass=r(d["mass"]), excluded=d["excluded"], special=d["special"], name="6mrr",
                  pme=dict(order=5, error_tol=5e-4, eps_r=1.0) if (pme and coulomb == "ewald") else None, **kw)


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

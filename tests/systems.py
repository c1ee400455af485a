"""Deterministic test systems shared by the parity tests: each Case can be realised as an oracle system
(oracle/pyoracle.py, the checker) and as a product System (molly_jl_amd, the thing under test)."""
import math

import numpy as np


CUT = {"none": 0, "distance": 1, "shifted_potential": 2, "shifted_force": 3, "cubic_spline": 4, "polynomial": 5}


def pme_mesh(box, alpha, error_tol=0.0005):
    """pme_params (ewald.jl:479-482): mesh points per axis, max(ceil(2 α L / (3 tol^0.2)), 6)"""
    return tuple(max(int(np.ceil(2.0 * alpha * float(L) / (3.0 * error_tol ** 0.2))), 6) for L in box)


class Case:
    def __init__(self, coords, box, lj=None, coul=None, r_list=math.inf, rebuild_every=10, velocities=None, charge=None,
                 sigma=None, eps=None, mass=None, excluded=None, special=None, bonds=None, angles=None, torsions=None,
                 ewald_excl=None, name="case", pme=None, triclinic=None):
        """lj: None | dict(cutoff=(kind, rc[, ra]), weight_special=1.0)
        coul: None | dict(kind="plain"|"rf"|"ewald", cutoff=(kind, rc[, ra]) (plain), rc=…, eps_rf=78.3, tol=5e-4,
                          approx=True, weight_special=1.0)"""
        self.coords = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
        self.n = len(self.coords)
        self.box = np.broadcast_to(np.asarray(box, dtype=np.float64), (3,)).copy()
        self.lj, self.coul = lj, coul
        self.r_list, self.rebuild_every = r_list, rebuild_every
        self.velocities = None if velocities is None else np.asarray(velocities, dtype=np.float64)
        self.charge, self.sigma, self.eps, self.mass = charge, sigma, eps, mass
        self.excluded, self.special = excluded, special
        self.bonds, self.angles, self.torsions, self.ewald_excl = bonds, angles, torsions, ewald_excl
        self.name = name
        self.triclinic = triclinic   # None | dict(basis=3x3, approx_images=True): TriclinicBoundary; `box` = the basis' diagonal
        self.pme = pme       # None | dict(order=5, error_tol=5e-4, eps_r=1.0[, mesh=(nx, ny, nz)]): general interaction PME (needs coul kind "ewald")

    def pme_params(self, dtype):
        """order, mesh, ϵr of the PME general interaction as Molly's constructor derives them (ewald.jl:361-372, 479-482)"""
        if self.pme is None:
            return None
        alpha = self.inter_dict(dtype)["ewald_alpha"]
        mesh = self.pme.get("mesh") or pme_mesh(self.box, alpha, self.pme.get("error_tol", self.coul.get("tol", 5e-4)))
        return dict(order=self.pme.get("order", 5), mesh=tuple(int(v) for v in mesh), eps_r=self.pme.get("eps_r", 1.0))

    # -- interaction dict for the oracle (field names of mhip_interactions) ---------------------------------
    def inter_dict(self, dtype):
        T = np.dtype(dtype).type
        d = {}
        if self.lj is not None:
            c = self.lj.get("cutoff", ("none",))
            d.update(lj_enabled=1, lj_cutoff_kind=CUT[c[0]], lj_rc=c[1] if len(c) > 1 else 0.0,
                     lj_ra=c[2] if len(c) > 2 else 0.0, lj_weight_special=self.lj.get("weight_special", 1.0))
        if self.coul is not None:
            k = self.coul["kind"]
            d["coul_weight_special"] = self.coul.get("weight_special", 1.0)
            if k == "plain":
                c = self.coul.get("cutoff", ("none",))
                d.update(coul_kind=1, coul_cutoff_kind=CUT[c[0]], coul_rc=c[1] if len(c) > 1 else 0.0, coul_ra=c[2] if len(c) > 2 else 0.0)
            elif k == "rf":
                d.update(coul_kind=2, coul_rc=self.coul["rc"], rf_dielectric=self.coul.get("eps_rf", 78.3))
            elif k == "ewald":
                rc, tol = self.coul["rc"], self.coul.get("tol", 5e-4)
                alpha = float((T(1) / T(rc)) * np.sqrt(-np.log(T(2) * T(tol))))
                d.update(coul_kind=3, coul_rc=rc, ewald_alpha=alpha, ewald_approx_erfc=int(self.coul.get("approx", True)))
        return d

    def _tors(self):
        if self.torsions is None:
            return None
        t = dict(self.torsions)
        return t

    def oracle(self, dtype=np.float64, coords=None, velocities=None):
        from oracle import pyoracle as orc   # only where a checker is asked for: bench.py builds its systems from this module too
        return orc.OracleSystem(self.coords if coords is None else coords, self.box, self.inter_dict(dtype), dtype=dtype,
                                velocities=self.velocities if velocities is None else velocities,
                                charge=self.charge, sigma=self.sigma, eps=self.eps, mass=self.mass, r_list=self.r_list,
                                rebuild_every=self.rebuild_every, excluded=self.excluded, special=self.special,
                                bonds=self.bonds, angles=self.angles, torsions=self._tors(), ewald_excl=self.ewald_excl,
                                pme=self.pme_params(dtype), triclinic=self.triclinic)

    def system(self, m, dtype=np.float32, coords=None, velocities=None):
        """Product System with the reference-style constructors (m = the molly_jl_amd module)."""
        def cutoff(c):
            k = c[0]
            return {"none": lambda: m.NoCutoff(), "distance": lambda: m.DistanceCutoff(c[1]),
                    "shifted_potential": lambda: m.ShiftedPotentialCutoff(c[1]), "shifted_force": lambda: m.ShiftedForceCutoff(c[1]),
                    "cubic_spline": lambda: m.CubicSplineCutoff(c[2], c[1]), "polynomial": lambda: m.PolynomialCutoff(c[2], c[1])}[k]()
        use_nl = math.isfinite(self.r_list)
        inters = []
        if self.lj is not None:
            inters.append(m.LennardJones(cutoff=cutoff(self.lj.get("cutoff", ("none",))), use_neighbors=use_nl,
                                         weight_special=self.lj.get("weight_special", 1.0)))
        if self.coul is not None:
            k, w = self.coul["kind"], self.coul.get("weight_special", 1.0)
            if k == "plain":
                inters.append(m.Coulomb(cutoff=cutoff(self.coul.get("cutoff", ("none",))), use_neighbors=use_nl, weight_special=w))
            elif k == "rf":
                inters.append(m.CoulombReactionField(dist_cutoff=self.coul["rc"], solvent_dielectric=self.coul.get("eps_rf", 78.3),
                                                     use_neighbors=use_nl, weight_special=w))
            else:
                inters.append(m.CoulombEwald(dist_cutoff=self.coul["rc"], error_tol=self.coul.get("tol", 5e-4), use_neighbors=use_nl,
                                             weight_special=w, approximate_erfc=self.coul.get("approx", True), dtype=dtype))
        sils = []
        if self.bonds is not None:
            sils.append(m.HarmonicBonds(self.bonds["i"], self.bonds["j"], self.bonds["k"], self.bonds["r0"]))
        if self.angles is not None:
            sils.append(m.HarmonicAngles(self.angles["i"], self.angles["j"], self.angles["k"], self.angles["kth"], self.angles["th0"]))
        if self.torsions is not None:
            t = self.torsions
            sils.append(m.PeriodicTorsions(t["i"], t["j"], t["k"], t["l"], t["periodicity"], t["phase"], t["k0"]))
        if self.ewald_excl is not None:
            e = np.asarray(self.ewald_excl).reshape(-1, 2)
            sils.append(m.EwaldExclusions(e[:, 0], e[:, 1]))
        nf = m.GPUNeighborFinder(dist_cutoff=self.r_list, excluded_pairs=self.excluded, special_pairs=self.special,
                                 n_steps=self.rebuild_every) if use_nl or self.excluded is not None or self.special is not None else None
        gis = []
        if self.pme is not None:
            gis.append(m.PME(self.coul["rc"], boundary=m.CubicBoundary(*self.box), error_tol=self.pme.get("error_tol", self.coul.get("tol", 5e-4)),
                             order=self.pme.get("order", 5), ϵr=self.pme.get("eps_r", 1.0), dtype=dtype))
            gis[-1].mesh_dims = self.pme_params(dtype)["mesh"]      # the oracle and the product always see the same mesh
        boundary = m.CubicBoundary(*self.box) if self.triclinic is None else m.TriclinicBoundary(*np.asarray(self.triclinic["basis"], dtype=np.float64).reshape(3, 3),
                                                                                                  approx_images=self.triclinic.get("approx_images", True))
        return m.System(coords=self.coords if coords is None else coords, boundary=boundary,
                        velocities=self.velocities if velocities is None else velocities, pairwise_inters=tuple(inters),
                        specific_inter_lists=tuple(sils), neighbor_finder=nf, dtype=dtype, charge=self.charge,
                        sigma=self.sigma, eps=self.eps, mass=self.mass, general_inters=tuple(gis))


# ---- SURVEY §8(d) synthetic LJ fluid (argon at 1400 kg/m³, benchmark/benchmark_gpu_tiles.jl:18-25) ------
ARGON = dict(sigma=0.34, eps=0.997, mass=39.948)
LJ_SPACING = 0.36183   # nm → ρ = 21.105 nm⁻³


def lj_fluid(n_side, seed=2, temperature=85.0, jitter=0.02, r_cut=1.0, r_list=1.2, rebuild_every=10, dtype=np.float32):
    """n_side³ atoms on a jittered simple-cubic lattice, Maxwell-Boltzmann velocities with CM removed.
    Coordinates are rounded to `dtype` so that every precision sees identical inputs."""
    n = n_side ** 3
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    box = n_side * LJ_SPACING
    x = (g + 0.5) * LJ_SPACING + rng.uniform(-jitter, jitter, (n, 3))
    x = x - np.floor(x / box) * box
    x = x.astype(dtype).astype(np.float64)
    x = np.where(x >= np.float64(dtype(box)), 0.0, x)
    rngv = np.random.default_rng(seed + 1)
    v = rngv.normal(size=(n, 3)) * math.sqrt(8.314462618e-3 * temperature / ARGON["mass"])
    v -= v.mean(axis=0)
    v = v.astype(dtype).astype(np.float64)
    return Case(x, float(dtype(box)), lj=dict(cutoff=("distance", r_cut)), r_list=r_list, rebuild_every=rebuild_every, velocities=v,
                sigma=np.full(n, ARGON["sigma"]), eps=np.full(n, ARGON["eps"]), mass=np.full(n, ARGON["mass"]),
                name=f"lj{n}")


def charged_fluid(n_side, coul, seed=5, spacing=0.31, r_list=1.2, dtype=np.float32, with_exceptions=True, stable=False, pme=None, box_scale=(1.0, 1.0, 1.0)):
    """A water-like-density mixed LJ + Coulomb fluid with per-atom σ, ϵ, q (two species + some LJ-less
    'hydrogens' with ϵ = 0) and random excluded / special pairs between close atoms.  The ϵ = 0 species exercises the
    LJZeroShortcut in force tests but, being free point charges, collapses onto opposite charges within a few dozen steps;
    dynamics tests pass stable=True, which gives it a small repulsive core."""
    n = n_side ** 3
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    box = n_side * spacing
    x = (g + 0.5) * spacing + rng.uniform(-0.04, 0.04, (n, 3))
    x = x - np.floor(x / box) * box
    x = x.astype(dtype).astype(np.float64)
    x = np.where(x >= np.float64(dtype(box)), 0.0, x)
    kind = rng.integers(0, 3, n)
    sigma = np.choose(kind, [0.315, 0.25, 0.2 if stable else 0.1]); eps = np.choose(kind, [0.65, 0.3, 0.2 if stable else 0.0])
    q = np.choose(kind, [-0.8, 0.35, 0.45]) * rng.uniform(0.9, 1.1, n)
    q -= q.mean()
    excluded = special = None
    if with_exceptions:
        # pairs of lattice neighbours: (i, i+1) excluded for a third of the atoms, (i, i+2) special for another third
        idx = np.arange(n - 2)
        excluded = np.stack([idx[idx % 3 == 0], idx[idx % 3 == 0] + 1], 1)
        special = np.stack([idx[idx % 3 == 1], idx[idx % 3 == 1] + 2], 1)
    v = rng.normal(size=(n, 3)) * 0.3
    v -= v.mean(axis=0)
    boxv = np.array([float(dtype(box * sc)) for sc in box_scale])     # box_scale > 1 stretches the box (orthorhombic cases), atoms stay put
    return Case(x, boxv, lj=dict(cutoff=("distance", 1.0), weight_special=0.5), coul=coul, r_list=r_list,
                velocities=v.astype(dtype).astype(np.float64), charge=q.astype(dtype).astype(np.float64),
                sigma=sigma, eps=eps, mass=np.choose(kind, [15.999, 12.011, 1.008]), excluded=excluded, special=special,
                name=f"charged{n}", pme=pme)


def pair_keys(i, j):
    """sorted uint64 keys lo << 32 | hi of a pair list: set comparison of tens of millions of pairs in seconds"""
    lo, hi = np.minimum(i, j).astype(np.uint64), np.maximum(i, j).astype(np.uint64)
    k = (lo << np.uint64(32)) | hi
    k.sort()
    return k


def export_keys(pkg, s):
    """the engine's neighbour list (mhip_export_neighbors) as sorted pair keys, + the special flags' count"""
    import ctypes as C
    L = pkg.lib()
    n = C.c_int64(0)
    s._check(L.mhip_export_neighbors(s.engine(), None, None, None, 0, C.byref(n)))
    i = np.empty(n.value, np.int32); j = np.empty(n.value, np.int32); sp = np.empty(n.value, np.uint8)
    s._check(L.mhip_export_neighbors(s.engine(), s._ptr(i), s._ptr(j), s._ptr(sp), n.value, C.byref(n)))
    return pair_keys(i, j), int(sp.sum())


def sorted_pairs(i, j, sp):
    lo, hi = np.minimum(i, j).astype(np.int64), np.maximum(i, j).astype(np.int64)
    order = np.lexsort((hi, lo))
    return lo[order], hi[order], np.asarray(sp, dtype=np.uint8)[order]


def fp32_force_tolerance(case, coords=None, rel=4e-5):
    """Per-atom tolerance for fp32 forces against the fp64 oracle: rel·Σ_j‖f_ij‖ plus the force jump of
    any pair sitting within 2e-6 (relative) of a hard cutoff, where fp32 may legitimately flip `r <= rc`.
    Calibration: the reference's own arithmetic evaluated in fp32 (oracle float instantiation, correctly
    rounded libm) measures max 2.9e-5·Σ_j‖f_ij‖ and relative RMS 4.9e-6 against fp64 on the 262 144-atom
    LJ fluid; the bar for the HIP path is max 4e-5 and relative RMS 1e-5."""
    o = case.oracle(np.float64, coords=coords)
    nl = o.neighbors("cell") if math.isfinite(case.r_list) else None
    scale, jump = o.force_scale(nl)
    o.pair_force_scale = scale   # Σ_j‖f_ij‖ per atom, kept for property checks
    return rel * scale + 1.01 * jump + 1e-6, o, nl


def rel_rms(err, f_ref):
    return float(np.sqrt((err ** 2).sum() / (np.linalg.norm(f_ref, axis=1) ** 2).sum()))


def fp32_reference_rms(case, f_ref64, coords=None, specific=False, nthreads=8):
    """Relative RMS force error of the REFERENCE's arithmetic evaluated in fp32 (oracle float instantiation, correctly
    rounded libm, list order summation) against fp64 on the same inputs: the yardstick for the fp32 HIP path, which
    must be no worse than 1.5x this (or 5e-6, whichever is larger)."""
    o32 = case.oracle(np.float32, coords=coords)
    nl32 = o32.neighbors("cell", nthreads=nthreads) if math.isfinite(case.r_list) else None
    f32 = o32.forces(nl32, nthreads=1, specific=specific).astype(np.float64)
    return rel_rms(np.linalg.norm(f32 - f_ref64, axis=1), f_ref64)

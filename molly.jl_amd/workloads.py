"""Synthetic workloads of SURVEY §8(d) and the description of a system they share with the tests: a `Case` holds coordinates, box,
per-atom parameters, interactions and topology as plain numpy data and realises itself as a product `System` (api.py).  The generators
are the benchmark's inputs (bench.py) and the parity tests' (tests/systems.py, which adds the checker's side of a case)."""
import math
import os

import numpy as np


CUT = {"none": 0, "distance": 1, "shifted_potential": 2, "shifted_force": 3, "cubic_spline": 4, "polynomial": 5}


def pme_mesh(box, alpha, error_tol=0.0005):
    """pme_params (ewald.jl:479-482): mesh points per axis, max(ceil(2 α L / (3 tol^0.2)), 6)"""
    return tuple(max(int(np.ceil(2.0 * alpha * float(L) / (3.0 * error_tol ** 0.2))), 6) for L in box)


class Case:
    def __init__(self, coords, box, lj=None, coul=None, r_list=math.inf, rebuild_every=10, velocities=None, charge=None,
                 sigma=None, eps=None, mass=None, excluded=None, special=None, bonds=None, angles=None, torsions=None,
                 ewald_excl=None, name="case", pme=None, triclinic=None, lam=None):
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
        self.lam = lam       # None | per-atom λ (Atom.λ, types.jl:466-475): the unsoftened interactions read it in the LJZeroShortcut only (mixing.jl:7-11)
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
                        sigma=self.sigma, eps=self.eps, mass=self.mass, general_inters=tuple(gis), lam=self.lam)


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


def argon_random(n_atoms=4096, box_multiplier=1.0, seed=42, r_cut=1.2, dtype=np.float32):
    """The system of the reference's own GPU benchmark (benchmark/benchmark_gpu_tiles.jl:13-40): n argon atoms at UNIFORMLY RANDOM positions in a
    cubic box of 1400 kg/m³ (× box_multiplier per side, at least 2.5 r_cut), LennardJones with DistanceCutoff(1.2 nm), zero velocities; the list
    radius is the cutoff (GPUNeighborFinder(dist_cutoff = r_cut)).  dense_f32: multiplier 1 (5.79 nm), sparse_f32: multiplier 4."""
    vol = n_atoms * ARGON["mass"] / (1400.0 * 6.02214076e23) * 1e24          # nm³ (g/mol ÷ (kg/m³ · mol⁻¹) = 1e-3 m³ → ·1e27 nm³/m³)
    box = max(vol ** (1.0 / 3.0) * box_multiplier, 2.5 * r_cut)
    rng = np.random.default_rng(seed)
    x = (rng.random((n_atoms, 3)) * box).astype(dtype).astype(np.float64)
    x = np.where(x >= np.float64(dtype(box)), 0.0, x)
    return Case(x, float(dtype(box)), lj=dict(cutoff=("distance", r_cut)), r_list=r_cut, rebuild_every=10, velocities=np.zeros((n_atoms, 3)),
                sigma=np.full(n_atoms, ARGON["sigma"]), eps=np.full(n_atoms, ARGON["eps"]), mass=np.full(n_atoms, ARGON["mass"]),
                name=f"argon{n_atoms}_x{box_multiplier:g}")


MEMLIMIT = dict(volume_per_atom=0.013, sigma=0.001, eps=0.1, mass=10.0, r_cut=1.0, dt=0.0001, n_steps=100, n_steps_reorder=25)


def memlimit_box(n_atoms):
    """box side of the reference's "Testing GPU memory limits" recipe (docs/src/examples.md:975-981): V = n_atoms * 0.013f0 nm³, cbrt(V), both in Float32"""
    return float(np.cbrt(np.float32(n_atoms) * np.float32(MEMLIMIT["volume_per_atom"]), dtype=np.float32))


def memlimit_fluid(n_atoms, seed=7, dtype=np.float32):
    """The system of docs/src/examples.md:969-1000: n atoms (mass 10, σ 0.001 nm, ϵ 0.1 kJ/mol, no charge) at UNIFORMLY RANDOM positions at 76.9 atoms/nm³,
    LennardJones(DistanceCutoff(1.0)) over a GPUNeighborFinder(dist_cutoff = 1.0) (n_steps_reorder = 25, neighbors.jl:327), zero velocities,
    VelocityVerlet(dt = 0.1 fs, remove_CM_motion = false).  bench.py --workload memlimit generates the same thing on the device for sizes a host array would not suit."""
    box = memlimit_box(n_atoms)
    rng = np.random.default_rng(seed)
    x = (rng.random((n_atoms, 3), dtype=np.float32) * np.float32(box)).astype(dtype).astype(np.float64)
    x = np.where(x >= np.float64(dtype(box)), 0.0, x)
    P = MEMLIMIT
    return Case(x, float(dtype(box)), lj=dict(cutoff=("distance", P["r_cut"])), r_list=P["r_cut"], rebuild_every=P["n_steps_reorder"], velocities=np.zeros((n_atoms, 3)),
                sigma=np.full(n_atoms, P["sigma"]), eps=np.full(n_atoms, P["eps"]), mass=np.full(n_atoms, P["mass"]), name=f"memlimit{n_atoms}")


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


# ---- 6mrr (15 954 atoms, Amber ff99SB-ILDN + TIP3P): the reference's data/6mrr_equil.pdb + force-field XML as flat arrays (coordinates, box,
# 300 K velocities, charges, σ, ϵ, masses, bonded terms, exclusions), written by tools/param_6mrr.py into the package's own data directory.
# The OpenMM outputs the parity tests compare with live in tests/golden/6mrr.npz and are never read from here.
PROTEIN_6MRR_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "6mrr_system.npz")
_npz_cache = {}


def protein_6mrr_data(path=None):
    path = path or PROTEIN_6MRR_NPZ
    if path not in _npz_cache:
        _npz_cache[path] = dict(np.load(path))
    return _npz_cache[path]


def protein_6mrr(coulomb="rf", dtype=np.float64, bonded=True, lj=True, which_bonded=("bonds", "angles", "proper", "improper"), r_list=1.2,
                 approx_erfc=True, rebuild_every=10, pme=False, path=None):
    """coulomb: None | "rf" (CoulombReactionField rc 1.0, ε 78.3 — OpenMM CutoffPeriodic) | "ewald" (CoulombEwald rc 1.0,
    tol 5e-4 + EwaldExclusion list), as setup.jl:1852-1913 wires them for nonbonded_method :cutoff / :pme."""
    d = protein_6mrr_data(path)
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
    return Case(coords, box, lj=dict(cutoff=("distance", 1.0), weight_special=float(d["weight_14_lj"])) if lj else None, coul=coul,
                r_list=r_list, rebuild_every=rebuild_every, velocities=r(d["velocities_300K"]), charge=r(d["charge"]), sigma=r(d["sigma"]),
                eps=r(d["eps"]), mass=r(d["mass"]), excluded=d["excluded"], special=d["special"], name="6mrr",
                pme=dict(order=5, error_tol=5e-4, eps_r=1.0) if (pme and coulomb == "ewald") else None, **kw)


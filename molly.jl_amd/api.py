"""Host-side mirror of the slice of Molly.jl's API that fronts the nonbonded hot path.

Julia is not available in this image, so this module stands where `MollyHIPExt.jl` stands in
a Julia session: same names, argument meaning and error behaviour as the reference, calling the same C ABI.
Indices are 0-based here (Python), 1-based in Julia.

Reference interfaces mirrored (Molly.jl v0.23.3):
  System                      src/types.jl:795-979
  Atom                        src/types.jl:466-475
  CubicBoundary               src/spatial.jl:40
  NoCutoff … PolynomialCutoff src/cutoffs.jl:53-253
  LennardJones                src/interactions/lennard_jones.jl:25-47
  Coulomb / CoulombReactionField / CoulombEwald   src/interactions/coulomb.jl:32, 698, 1320
  DistanceNeighborFinder / GPUNeighborFinder      src/neighbors.jl:104-115, 376-388
  NeighborList                src/types.jl:611-654
  forces / potential_energy / kinetic_energy / temperature   src/force.jl:670-720, src/energy.jl:86-248
  find_neighbors              src/neighbors.jl:390-423
  VelocityVerlet, simulate!   src/simulators.jl:287-300, 547-668
  remove_CM_motion!           src/spatial.jl:901-929
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import MollyHipError

COULOMB_CONST = 138.93545764   # coulomb.jl:16, kJ mol^-1 nm e^-2
BOLTZMANN = 8.314462618e-3     # kJ mol^-1 K^-1 (setup.jl:2007)


# ---- cutoffs (cutoffs.jl) ---------------------------------------------------------------------------
@dataclass(frozen=True)
class NoCutoff:
    kind = _lib.CUTOFF_NONE
    dist_cutoff: float = 0.0
    dist_activation: float = 0.0


@dataclass(frozen=True)
class DistanceCutoff:
    dist_cutoff: float
    kind = _lib.CUTOFF_DISTANCE
    dist_activation: float = 0.0


@dataclass(frozen=True)
class ShiftedPotentialCutoff:
    dist_cutoff: float
    kind = _lib.CUTOFF_SHIFTED_POTENTIAL
    dist_activation: float = 0.0


@dataclass(frozen=True)
class ShiftedForceCutoff:
    dist_cutoff: float
    kind = _lib.CUTOFF_SHIFTED_FORCE
    dist_activation: float = 0.0


class _TwoRadiusCutoff:
    def __init__(self, dist_activation, dist_cutoff):
        if dist_cutoff <= dist_activation:   # cutoffs.jl:180-183, 234-237
            raise ValueError(f"the cutoff radius {dist_cutoff} must be larger than the activation radius {dist_activation}")
        self.dist_activation, self.dist_cutoff = float(dist_activation), float(dist_cutoff)


class CubicSplineCutoff(_TwoRadiusCutoff):
    kind = _lib.CUTOFF_CUBIC_SPLINE


class PolynomialCutoff(_TwoRadiusCutoff):
    kind = _lib.CUTOFF_POLYNOMIAL


# ---- pairwise interactions ----------------------------------------------------------------------------
@dataclass
class LennardJones:
    """LennardJones(; cutoff, use_neighbors, weight_special) with Lorentz σ / geometric ϵ mixing and the
    LJZeroShortcut (the defaults, lennard_jones.jl:29-34)."""
    cutoff: object = field(default_factory=NoCutoff)
    use_neighbors: bool = False
    weight_special: float = 1.0


@dataclass
class Coulomb:
    cutoff: object = field(default_factory=NoCutoff)
    use_neighbors: bool = False
    weight_special: float = 1.0
    coulomb_const: float = COULOMB_CONST


@dataclass
class CoulombReactionField:
    dist_cutoff: float
    solvent_dielectric: float = 78.3   # coulomb.jl:705 crf_solvent_dielectric
    use_neighbors: bool = False
    weight_special: float = 1.0
    coulomb_const: float = COULOMB_CONST


class CoulombEwald:
    def __init__(self, dist_cutoff, error_tol=0.0005, use_neighbors=False, weight_special=1.0,
                 coulomb_const=COULOMB_CONST, approximate_erfc=True, dtype=np.float64):
        self.dist_cutoff, self.error_tol = float(dist_cutoff), float(error_tol)
        self.use_neighbors, self.weight_special, self.coulomb_const = use_neighbors, float(weight_special), float(coulomb_const)
        self.approximate_erfc = bool(approximate_erfc)
        T = np.dtype(dtype).type
        # α = inv(dist_cutoff) * sqrt(-log(2 * error_tol)) evaluated in T (coulomb.jl:1332)
        self.α = float((T(1) / T(dist_cutoff)) * np.sqrt(-np.log(T(2) * T(error_tol))))


def use_neighbors(inter):
    return inter.use_neighbors


class PME:
    """PME(dist_cutoff, atoms, boundary; error_tol=0.0005, order=5, ϵr=1.0): reciprocal-space part of the particle-mesh Ewald
    sum, a general interaction (ewald.jl:361-421).  α and the mesh are derived as the reference does (:368-369, :479-482)."""

    def __init__(self, dist_cutoff, atoms=None, boundary=None, error_tol=0.0005, order=5, ϵr=1.0, dtype=np.float64):
        if boundary is None:
            raise ValueError("PME needs the boundary")
        T = np.dtype(dtype).type
        self.dist_cutoff, self.error_tol, self.order, self.ϵr = float(dist_cutoff), float(error_tol), int(order), float(ϵr)
        self.α = float((T(1) / T(dist_cutoff)) * np.sqrt(-np.log(T(2) * T(error_tol))))
        sides = boundary.side_lengths if isinstance(boundary, (CubicBoundary, TriclinicBoundary)) else tuple(boundary)      # box_sides(boundary), spatial.jl:357-360
        tol = T(error_tol)
        self.mesh_dims = tuple(max(int(np.ceil(T(2) * T(self.α) * T(L) / (T(3) * tol ** T(0.2)))), 6) for L in sides)   # pme_params


# ---- specific interaction lists (SoA form of InteractionList{2,3,4}Atoms, types.jl:236-420) ---------
@dataclass
class HarmonicBonds:
    i: np.ndarray; j: np.ndarray; k: np.ndarray; r0: np.ndarray


@dataclass
class HarmonicAngles:
    i: np.ndarray; j: np.ndarray; k: np.ndarray; kθ: np.ndarray; θ0: np.ndarray


@dataclass
class PeriodicTorsions:
    """One entry per (torsion, Fourier term); proper and improper torsions may share one list."""
    i: np.ndarray; j: np.ndarray; k: np.ndarray; l: np.ndarray
    periodicity: np.ndarray; phase: np.ndarray; k0: np.ndarray


@dataclass
class EwaldExclusions:
    i: np.ndarray; j: np.ndarray


# ---- boundary / atoms / neighbour finder ------------------------------------------------------------
class CubicBoundary:
    def __init__(self, x, y=None, z=None):
        if y is None:
            sl = np.broadcast_to(np.asarray(x, dtype=np.float64), (3,)).copy()
        else:
            sl = np.array([x, y, z], dtype=np.float64)
        if np.any(sl <= 0):   # spatial.jl:46-48
            raise ValueError("CubicBoundary side lengths must be positive")
        self.side_lengths = sl


class TriclinicBoundary:
    """TriclinicBoundary(v1, v2, v3; approx_images=true) (spatial.jl:131-220): v1 along x, v2 in the xy plane, v3.z > 0.  The engine
    supports it on a single GPU, PME included (include/mollyhip.h, mhip_set_triclinic)."""

    def __init__(self, v1, v2, v3, approx_images=True):
        bv = np.array([v1, v2, v3], dtype=np.float64).reshape(3, 3)
        if not (bv[0, 0] > 0) or bv[0, 1] != 0 or bv[0, 2] != 0:   # spatial.jl:173-177
            raise ValueError(f"first basis vector must be along the x-axis (no y or z component) and have a positive x component, got {bv[0]}")
        if not (bv[1, 1] > 0) or bv[1, 2] != 0:                     # :178-182
            raise ValueError(f"second basis vector must be in the xy plane (no z component) and have a positive y component, got {bv[1]}")
        if not (bv[2, 2] > 0):                                      # :183-186
            raise ValueError(f"third basis vector must have a positive z component, got {bv[2]}")
        self.basis_vectors = bv
        self.approx_images = bool(approx_images)
        self.side_lengths = np.array([bv[0, 0], bv[1, 1], bv[2, 2]])     # box_volume = v1.x · v2.y · v3.z; the engine's config box
        self.reciprocal_size = 1.0 / self.side_lengths


@dataclass
class Atom:
    index: int = 0
    atom_type: int = 0
    mass: float = 1.0
    charge: float = 0.0
    σ: float = 0.0
    ϵ: float = 0.0
    λ: float = 1.0


class NeighborList:
    """n entries (i, j, special), i < j, 0-based."""
    def __init__(self, i, j, special):
        self.i, self.j, self.special = i, j, special
        self.n = len(i)

    @property
    def list(self):
        return list(zip(self.i.tolist(), self.j.tolist(), self.special.astype(bool).tolist()))


def _pairs_from(arg, name):
    """Accepts a dense boolean matrix (as Molly's eligible/special) or an (m,2) array of pairs."""
    if arg is None:
        return np.zeros((0, 2), np.int32)
    a = np.asarray(arg)
    if a.dtype == bool and a.ndim == 2 and a.shape[0] == a.shape[1]:
        ii, jj = np.nonzero(np.triu(a, 1) | np.triu(a.T, 1))
        return np.stack([ii, jj], 1).astype(np.int32)
    a = a.reshape(-1, 2).astype(np.int64)
    if len(a) and (a.min() < 0):
        raise ValueError(f"{name}: negative atom index")
    lo, hi = np.minimum(a[:, 0], a[:, 1]), np.maximum(a[:, 0], a[:, 1])
    keep = lo != hi
    # normalize_pairs: i<j, sorted, unique (neighbors.jl:171-195)
    u = np.unique(np.stack([lo[keep], hi[keep]], 1), axis=0) if keep.any() else np.zeros((0, 2), np.int64)
    return u.astype(np.int32)


class GPUNeighborFinder:
    """GPUNeighborFinder(; eligible, dist_cutoff, special, n_steps_reorder) — neighbors.jl:104-115, 320-361.
    `eligible`/`special` may be dense Bool matrices or sparse pair arrays (`excluded_pairs`, `special_pairs`)."""
    def __init__(self, dist_cutoff, eligible=None, special=None, excluded_pairs=None, special_pairs=None, n_steps=10):
        self.dist_cutoff = float(dist_cutoff)
        self.n_steps = int(n_steps)
        if eligible is not None:
            el = np.asarray(eligible, dtype=bool)
            excluded_pairs = np.argwhere(np.triu(~el | ~el.T, 1))
        self.excluded = _pairs_from(excluded_pairs, "excluded_pairs")
        self.special = _pairs_from(special if special is not None else special_pairs, "special_pairs")


DistanceNeighborFinder = GPUNeighborFinder       # same contract for this engine (neighbors.jl:376-388)
CellListMapNeighborFinder = GPUNeighborFinder    # neighbors.jl:543-590


class NoNeighborFinder:
    dist_cutoff = math.inf
    n_steps = 10
    excluded = np.zeros((0, 2), np.int32)
    special = np.zeros((0, 2), np.int32)


@dataclass
class VelocityVerlet:
    dt: float
    coupling: object = None
    remove_CM_motion: int = 1    # simulators.jl:293


@dataclass
class SteepestDescentMinimizer:
    """SteepestDescentMinimizer(; step_size=0.01 nm, max_steps=1000, tol=1000 kJ mol⁻¹ nm⁻¹) (simulators.jl:112-135): x += hn · F / max|F|, the step taken when
    the potential energy falls (hn · 6/5) and taken back otherwise (hn / 5), until max|F| < tol"""
    step_size: float = 0.01
    max_steps: int = 1000
    tol: float = 1000.0
    log_stream: object = None


@dataclass
class AndersenThermostat:
    """AndersenThermostat(temperature, coupling_const) (coupling.jl:188-211): every step each atom's velocity is re-drawn from the
    Maxwell-Boltzmann distribution with probability dt / coupling_const."""
    temperature: float
    coupling_const: float


class Langevin:
    """Langevin(; dt, temperature, friction, coupling=nothing, remove_CM_motion=1) — the Langevin middle integrator
    (simulators.jl:1065-1097): vel_scale = exp(−dt·friction), noise_scale = sqrt(1 − vel_scale²)."""

    def __init__(self, dt, temperature, friction, coupling=None, remove_CM_motion=1):
        self.dt, self.temperature, self.friction = float(dt), float(temperature), float(friction)
        self.coupling, self.remove_CM_motion = coupling, int(remove_CM_motion)
        self.vel_scale = float(np.exp(-self.dt * self.friction))
        self.noise_scale = float(np.sqrt(1.0 - self.vel_scale ** 2))


def _rng(rng):
    return rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)


def _rand_u64(rng):
    return int(rng.integers(0, 2 ** 64, dtype=np.uint64))          # rand(rng, UInt64)


# ---- System --------------------------------------------------------------------------------------------
class System:
    """System(; atoms, coords, boundary, velocities, pairwise_inters, specific_inter_lists, neighbor_finder)
    with array type "MI355X": every force/energy/integration call runs in libmollyhip.so."""

    def __init__(self, atoms=None, coords=None, boundary=None, velocities=None, pairwise_inters=(),
                 specific_inter_lists=(), neighbor_finder=None, dtype=np.float32, device_id=0,
                 charge=None, sigma=None, eps=None, mass=None, general_inters=(), lam=None):
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        T = self.dtype
        if coords is None or boundary is None:
            raise ValueError("coords and boundary are required")
        self.coords = np.ascontiguousarray(coords, dtype=T).reshape(-1, 3).copy()
        n = len(self.coords)
        if atoms is not None:
            if len(atoms) != n:   # types.jl:914-916
                raise ValueError(f"there are {len(atoms)} atoms but {n} coordinates")
            charge = [a.charge for a in atoms]; sigma = [a.σ for a in atoms]; eps = [a.ϵ for a in atoms]
            mass = [a.mass for a in atoms]; lam = [a.λ for a in atoms]
            self.λ = np.ascontiguousarray(lam, dtype=T)
        else:
            self.λ = None if lam is None else np.ascontiguousarray(lam, dtype=T).reshape(n)      # (plain-array form of Atom.λ)
        arr = lambda a, d: np.full(n, d, T) if a is None else np.ascontiguousarray(a, dtype=T).reshape(n)
        self.charge, self.σ, self.ϵ, self.masses = arr(charge, 0), arr(sigma, 0), arr(eps, 0), arr(mass, 1)
        if velocities is not None and len(velocities) != n:
            raise ValueError(f"there are {n} coordinates but {len(velocities)} velocities")
        self.velocities = np.zeros((n, 3), T) if velocities is None else np.ascontiguousarray(velocities, dtype=T).reshape(n, 3).copy()
        self._ctx = None
        self._box_dirty = False
        self.boundary = boundary
        self.pairwise_inters = tuple(pairwise_inters)
        self.specific_inter_lists = tuple(specific_inter_lists)
        self.general_inters = tuple(general_inters)
        self.neighbor_finder = neighbor_finder if neighbor_finder is not None else NoNeighborFinder()
        self.device_id = device_id
        self.total_mass = float(self.masses.sum(dtype=np.float64))
        self._pushed_atoms = False

    def __len__(self):
        return len(self.coords)

    # `sys.boundary = …` on a live system (scale_coords!, spatial.jl:1202; the barostats' rejected moves, coupling.jl:930): the reference reads
    # sys.boundary at every force / energy call, so the engine's context follows the assignment (mhip_set_box) the next time state is pushed
    @property
    def boundary(self):
        return self._boundary

    @boundary.setter
    def boundary(self, b):
        b = b if isinstance(b, (CubicBoundary, TriclinicBoundary)) else CubicBoundary(b)
        old = getattr(self, "_boundary", None)
        if self._ctx is not None and old is not None:
            if type(b) is not type(old) or (isinstance(b, TriclinicBoundary) and b.approx_images != old.approx_images):
                raise MollyHipError(-6, "a live System keeps its kind of boundary (CubicBoundary / TriclinicBoundary and its image mode)")
            self._box_dirty = True
        self._boundary = b

    def _push_box(self):
        if not self._box_dirty or self._ctx is None:
            return
        b = self._boundary
        box = np.ascontiguousarray(b.side_lengths, dtype=np.float64)
        bv = np.ascontiguousarray(b.basis_vectors, dtype=np.float64) if isinstance(b, TriclinicBoundary) else None
        self._check(_lib.lib().mhip_set_box(self._ctx, self._ptr(box), self._ptr(bv)))
        self._box_dirty = False

    # -- interaction tuple → mhip_interactions ------------------------------------------------------------
    def interactions(self):
        it = _lib.Interactions()
        it.lj_weight_special = 1.0; it.coul_weight_special = 1.0; it.coul_ke = COULOMB_CONST; it.rf_dielectric = 1.0
        it.ewald_approx_erfc = 1
        n_coul = 0
        for inter in self.pairwise_inters:
            if isinstance(inter, LennardJones):
                if it.lj_enabled:
                    raise ValueError("only one LennardJones interaction is supported per System")
                it.lj_enabled = 1; it.lj_cutoff_kind = inter.cutoff.kind
                it.lj_rc = inter.cutoff.dist_cutoff; it.lj_ra = inter.cutoff.dist_activation
                it.lj_weight_special = inter.weight_special
            elif isinstance(inter, Coulomb):
                n_coul += 1
                it.coul_kind = _lib.COUL_PLAIN; it.coul_cutoff_kind = inter.cutoff.kind
                it.coul_rc = inter.cutoff.dist_cutoff; it.coul_ra = inter.cutoff.dist_activation
                it.coul_ke = inter.coulomb_const; it.coul_weight_special = inter.weight_special
            elif isinstance(inter, CoulombReactionField):
                n_coul += 1
                it.coul_kind = _lib.COUL_REACTION_FIELD; it.coul_rc = inter.dist_cutoff
                it.rf_dielectric = inter.solvent_dielectric; it.coul_ke = inter.coulomb_const
                it.coul_weight_special = inter.weight_special
            elif isinstance(inter, CoulombEwald):
                n_coul += 1
                it.coul_kind = _lib.COUL_EWALD_DIRECT; it.coul_rc = inter.dist_cutoff; it.ewald_alpha = inter.α
                it.ewald_approx_erfc = int(inter.approximate_erfc); it.coul_ke = inter.coulomb_const
                it.coul_weight_special = inter.weight_special
            else:
                raise MollyHipError(-6, f"pairwise interaction {type(inter).__name__} is outside the hot-path scope")
        if n_coul > 1:
            raise ValueError("only one Coulomb-type interaction is supported per System")
        return it

    def _uses_list(self):
        inters = self.pairwise_inters
        return any(use_neighbors(i) for i in inters) and math.isfinite(self.neighbor_finder.dist_cutoff)

    # -- engine context ----------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise MollyHipError(rc, _lib.lib().mhip_last_error(self._ctx).decode())

    def engine(self):
        if self._ctx is not None:
            return self._ctx
        L = _lib.lib()
        inters = self.pairwise_inters
        if inters and len({bool(use_neighbors(i)) for i in inters}) > 1:
            raise MollyHipError(-6, "mixing use_neighbors=true and false interactions is not supported")
        cfg = _lib.Config()
        cfg.precision = 32 if self.dtype == np.float32 else 64
        cfg.device_id = self.device_id
        cfg.n_atoms = len(self)
        for d in range(3):
            cfg.box[d] = self.boundary.side_lengths[d]; cfg.origin[d] = 0.0; cfg.periodic[d] = 1
        cfg.rebuild_every = self.neighbor_finder.n_steps
        cfg.r_list = self.neighbor_finder.dist_cutoff if self._uses_list() else math.inf
        cfg.inter = self.interactions()
        ctx = C.c_void_p()
        rc = L.mhip_create(C.byref(ctx), C.byref(cfg))
        if rc != 0:
            raise MollyHipError(rc, L.mhip_last_error(None).decode())
        self._ctx = ctx
        if isinstance(self.boundary, TriclinicBoundary):
            bv = np.ascontiguousarray(self.boundary.basis_vectors, dtype=np.float64)
            self._check(L.mhip_set_triclinic(ctx, self._ptr(bv), 1 if self.boundary.approx_images else 0))
        self._push_atoms()
        return ctx

    def _ptr(self, a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def _push_atoms(self):
        L = _lib.lib()
        self._check(L.mhip_set_atoms(self._ctx, self._ptr(self.charge), self._ptr(self.σ), self._ptr(self.ϵ),
                                     self._ptr(self.masses), self._ptr(self.λ), _lib.MEM_HOST))
        nf = self.neighbor_finder
        if self._uses_list() or len(nf.excluded) or len(nf.special):
            ex = np.ascontiguousarray(nf.excluded, dtype=np.int32).reshape(-1, 2)
            sp = np.ascontiguousarray(nf.special, dtype=np.int32).reshape(-1, 2)
            exi, exj = np.ascontiguousarray(ex[:, 0]), np.ascontiguousarray(ex[:, 1])
            spi, spj = np.ascontiguousarray(sp[:, 0]), np.ascontiguousarray(sp[:, 1])
            self._check(L.mhip_set_exceptions(self._ctx, self._ptr(exi), self._ptr(exj), len(exi), self._ptr(spi), self._ptr(spj), len(spi)))
        T = self.dtype
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        f = lambda a: np.ascontiguousarray(a, dtype=T)
        for sil in self.specific_inter_lists:
            if isinstance(sil, HarmonicBonds):
                a = [i32(sil.i), i32(sil.j), f(sil.k), f(sil.r0)]
                self._check(L.mhip_set_bonds(self._ctx, len(a[0]), *map(self._ptr, a)))
            elif isinstance(sil, HarmonicAngles):
                a = [i32(sil.i), i32(sil.j), i32(sil.k), f(sil.kθ), f(sil.θ0)]
                self._check(L.mhip_set_angles(self._ctx, len(a[0]), *map(self._ptr, a)))
            elif isinstance(sil, PeriodicTorsions):
                a = [i32(sil.i), i32(sil.j), i32(sil.k), i32(sil.l), i32(sil.periodicity), f(sil.phase), f(sil.k0)]
                self._check(L.mhip_set_torsions(self._ctx, len(a[0]), *map(self._ptr, a)))
            elif isinstance(sil, EwaldExclusions):
                a = [i32(sil.i), i32(sil.j)]
                self._check(L.mhip_set_ewald_exclusions(self._ctx, len(a[0]), *map(self._ptr, a)))
            else:
                raise MollyHipError(-6, f"specific interaction list {type(sil).__name__} is outside the hot-path scope")
        for gi in self.general_inters:
            if isinstance(gi, PME):
                mesh = (C.c_int32 * 3)(*gi.mesh_dims)
                self._check(L.mhip_set_pme(self._ctx, gi.order, mesh, gi.α, gi.ϵr))
            else:
                raise MollyHipError(-6, f"general interaction {type(gi).__name__} is outside the hot-path scope")

    def push_state(self, velocities=True):
        L = _lib.lib()
        self.engine()
        self._push_box()
        self._check(L.mhip_set_state(self._ctx, self._ptr(self.coords), self._ptr(self.velocities) if velocities else None, _lib.MEM_HOST))

    def pull_state(self):
        L = _lib.lib()
        self._check(L.mhip_get_state(self._ctx, self._ptr(self.coords), self._ptr(self.velocities), _lib.MEM_HOST))

    def stats(self):
        st = _lib.Stats()
        self._check(_lib.lib().mhip_get_stats(self.engine(), C.byref(st)))
        return st.as_dict()

    def close(self):
        if self._ctx is not None:
            _lib.lib().mhip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- free functions mirroring Molly's exports ---------------------------------------------------------
def forces(sys, step_n=0, pairwise=True, specific=True, general=True):
    """forces(sys; …): pairwise + specific + general interaction forces, (n,3) array (force.jl:670-720, 792-795)."""
    L = _lib.lib()
    sys.push_state(velocities=False)
    out = np.zeros((len(sys), 3), sys.dtype)
    if pairwise and sys.pairwise_inters:
        sys._check(L.mhip_forces(sys._ctx, step_n, 1, sys._ptr(out), None, _lib.MEM_HOST))
    if specific and sys.specific_inter_lists:
        sys._check(L.mhip_specific_forces(sys._ctx, 1, sys._ptr(out), _lib.MEM_HOST))
    if general and sys.general_inters:
        sys._check(L.mhip_general_forces(sys._ctx, 1, sys._ptr(out), _lib.MEM_HOST))
    return out


def virial(sys, step_n=0, pairwise=True, specific=True, general=True):
    """virial(sys): 3×3 tensor Σ r ⊗ f of the pairwise (over the neighbour pairs, force.jl:848-852), specific (force.jl:991-1060) and
    general (PME reciprocal space, ewald.jl:701-723, 925-927) interactions (energy.jl:116-131)."""
    L = _lib.lib()
    sys.push_state(velocities=False)
    v = np.zeros(9, np.float64)
    vp = v.ctypes.data_as(C.c_void_p)
    if pairwise and sys.pairwise_inters:
        out = np.zeros((len(sys), 3), sys.dtype)
        sys._check(L.mhip_forces(sys._ctx, step_n, 0, sys._ptr(out), vp, _lib.MEM_HOST))
    if specific and sys.specific_inter_lists:
        sys._check(L.mhip_specific_virial(sys._ctx, vp))
    if general and sys.general_inters:
        sys._check(L.mhip_general_virial(sys._ctx, vp))
    return v.reshape(3, 3)


def scalar_virial(sys, step_n=0):
    """scalar_virial(sys) = tr(virial(sys)) (energy.jl:148-151)"""
    return float(np.trace(virial(sys, step_n)))


def pressure(sys, step_n=0):
    """pressure(sys) = (2K + W) / V with K = ½ Σ m v ⊗ v (spatial.jl:930-982), in kJ mol⁻¹ nm⁻³ (1 kJ mol⁻¹ nm⁻³ = 16.6054 bar)"""
    w = virial(sys, step_n)
    v = np.asarray(sys.velocities, dtype=np.float64); m = np.asarray(sys.masses, dtype=np.float64)
    k = 0.5 * np.einsum("i,ia,ib->ab", m, v, v)
    return (2.0 * k + w) / float(np.prod(sys.boundary.side_lengths))


def scalar_pressure(sys, step_n=0):
    return float(np.trace(pressure(sys, step_n))) / 3.0


def potential_energy(sys, step_n=0, pairwise=True, specific=True, general=True):
    """potential_energy(sys; …) (energy.jl:207-248, 409-446)."""
    L = _lib.lib()
    sys.push_state(velocities=False)
    total = 0.0
    pe = C.c_double(0)
    if pairwise and sys.pairwise_inters:
        sys._check(L.mhip_potential_energy(sys._ctx, step_n, C.byref(pe)))
        total += pe.value
    if specific and sys.specific_inter_lists:
        sys._check(L.mhip_specific_potential_energy(sys._ctx, C.byref(pe)))
        total += pe.value
    if general and sys.general_inters:
        sys._check(L.mhip_general_potential_energy(sys._ctx, C.byref(pe)))
        total += pe.value
    return total


def kinetic_energy(sys):
    L = _lib.lib()
    sys.push_state(velocities=True)
    ke = C.c_double(0)
    sys._check(L.mhip_kinetic_energy(sys._ctx, C.byref(ke)))
    return ke.value


def temperature(sys):
    """T = 2 KE / (df k), df = 3N − 3 for a fully periodic box (energy.jl:158-175)."""
    return 2 * kinetic_energy(sys) / ((3 * len(sys) - 3) * BOLTZMANN)


def total_energy(sys):
    return kinetic_energy(sys) + potential_energy(sys)


def find_neighbors(sys, step_n=0):
    """find_neighbors(sys, sys.neighbor_finder) → NeighborList (neighbors.jl:390-423); None without a list."""
    if not sys._uses_list():
        return None
    L = _lib.lib()
    sys.push_state(velocities=False)
    sys._check(L.mhip_rebuild(sys._ctx, step_n))
    n = C.c_int64(0)
    sys._check(L.mhip_export_neighbors(sys._ctx, None, None, None, 0, C.byref(n)))
    i = np.empty(n.value, np.int32); j = np.empty(n.value, np.int32); sp = np.empty(n.value, np.uint8)
    if n.value:
        sys._check(L.mhip_export_neighbors(sys._ctx, sys._ptr(i), sys._ptr(j), sys._ptr(sp), n.value, C.byref(n)))
    return NeighborList(i, j, sp)


def optimize_launch_config(sys, n_passes=20):
    """optimize_cuda_launch_config!(sys) (src/cuda_config.jl:53-62, ext/MollyCUDAExt.jl:594-642): time the candidate workgroup shapes of the
    search / pair kernels on this system, keep the fastest; returns [(block_atoms, j_split, µs per force pass), …] (µs < 0: not launchable)."""
    L = _lib.lib()
    sys.push_state(velocities=False)
    tr = (_lib.LaunchTrial * 16)()
    n = C.c_int32(0)
    sys._check(L.mhip_optimize_launch_config(sys.engine(), int(n_passes), tr, 16, C.byref(n)))
    return [(t.block_atoms, t.j_split, float(t.us_per_pass)) for t in tr[: min(n.value, 16)]]


def set_launch_config(sys, block_atoms=0, j_split=0):
    """set_cuda_launch_config!(sys; …) / reset_cuda_launch_config!(sys) with the defaults (src/cuda_config.jl:17-47)."""
    sys._check(_lib.lib().mhip_set_launch_config(sys.engine(), int(block_atoms), int(j_split)))


def remove_CM_motion(sys):
    """remove_CM_motion!(sys) (spatial.jl:901-929)."""
    L = _lib.lib()
    sys.push_state(velocities=True)
    sys._check(L.mhip_remove_cm(sys._ctx))
    sys.pull_state()
    return sys


def _minimize(sys, sim, init_step=0):
    """simulate!(sys, ::SteepestDescentMinimizer) (simulators.jl:183-271; systems without constraints or virtual sites): every force and energy from the engine, the
    update x += hn · F / max|F| and the accept / reject rule in the system's number type on the host side, as the reference's loop broadcasts them"""
    T = sys.dtype.type
    log = (lambda *a: print(*a, file=sim.log_stream)) if sim.log_stream is not None else (lambda *a: None)
    sys.coords[:] = wrap_coords(sys.coords, sys.boundary)
    E = potential_energy(sys, init_step)
    log("Step", init_step, "- potential energy", E, "- max force N/A - N/A")
    hn = T(sim.step_size)
    for step_n in range(init_step + 1, init_step + sim.max_steps + 1):
        F = forces(sys, step_n)
        max_force = np.sqrt((F * F).sum(axis=1, dtype=sys.dtype)).max()               # maximum(norm.(F))
        coords_copy = sys.coords.copy()
        sys.coords[:] = wrap_coords(sys.coords + (hn * F / max_force).astype(sys.dtype), sys.boundary)
        E_trial = potential_energy(sys, step_n)
        if E_trial < E:
            hn = T(6) * hn / T(5)
            E = E_trial
            log("Step", step_n, "- potential energy", E_trial, "- max force", max_force, "- accepted")
        else:
            sys.coords[:] = coords_copy
            hn = hn / T(5)
            log("Step", step_n, "- potential energy", E_trial, "- max force", max_force, "- rejected")
        if max_force < sim.tol:
            break
    return sys


def simulate(sys, sim, n_steps=None, init_step=0, check_nans=False, rng=None):
    """simulate!(sys, sim, n_steps; init_step, rng) for VelocityVerlet (simulators.jl:547-668) and Langevin (:1099-1220), with
    coupling nothing, AndersenThermostat, MonteCarloBarostat or a tuple of the two.  The step loop runs on the device; with a barostat it is cut at the
    multiples of barostat.n_steps, where apply_coupling! runs on the host side of the boundary with the engine's potential energies (the README's GPU
    example: Langevin + MonteCarloBarostat).  Coordinates and velocities come back when the call returns.  rng: a numpy Generator or a seed; as in the
    reference it supplies the Philox key / counter words and the barostat's uniform numbers."""
    if isinstance(sim, SteepestDescentMinimizer):
        if init_step < 0:
            raise ValueError("init_step must be non-negative")
        return _minimize(sys, sim, init_step)
    if not isinstance(sim, (VelocityVerlet, Langevin)):
        raise MollyHipError(-6, f"simulator {type(sim).__name__} is outside the hot-path scope")
    if n_steps is None:
        raise TypeError("simulate: n_steps is required for VelocityVerlet and Langevin")
    couplings = sim.coupling if isinstance(sim.coupling, (tuple, list)) else (() if sim.coupling is None else (sim.coupling,))
    thermostat = barostat = None
    for c in couplings:
        if isinstance(c, AndersenThermostat) and thermostat is None:
            thermostat = c
        elif isinstance(c, MonteCarloBarostat) and barostat is None:
            barostat = c
        else:
            raise MollyHipError(-6, f"coupling {type(c).__name__} is outside the hot-path scope")
    if init_step < 0:   # check_simulate_inputs
        raise ValueError("init_step must be non-negative")
    L = _lib.lib()
    rng = _rng(rng)
    sys.push_state(velocities=True)
    if thermostat is not None:
        sys._check(L.mhip_set_andersen(sys._ctx, BOLTZMANN * float(thermostat.temperature), float(sim.dt) / float(thermostat.coupling_const), _rand_u64(rng)))
    try:
        if isinstance(sim, Langevin):
            key, ctr1 = _rand_u64(rng), _rand_u64(rng)             # simulators.jl:1149-1150
        first, last = init_step, init_step + n_steps
        while first < last:
            nxt = last if barostat is None else min(last, (first // barostat.n_steps + 1) * barostat.n_steps)
            if isinstance(sim, Langevin):                         # ctr1 advances by one per step (simulators.jl:1190)
                sys._check(L.mhip_langevin_run(sys._ctx, first, nxt - first, sim.dt, BOLTZMANN * sim.temperature, sim.friction,
                                               int(sim.remove_CM_motion), key, (ctr1 + (first - init_step)) % 2 ** 64))
            else:
                sys._check(L.mhip_vv_run(sys._ctx, first, nxt - first, float(sim.dt), int(sim.remove_CM_motion)))
            if barostat is not None and nxt % barostat.n_steps == 0:      # apply_coupling! at the end of step nxt (simulators.jl:1192)
                sys.pull_state()
                _apply_mc_barostat(sys, barostat, nxt, rng)
                sys.push_state(velocities=True)                   # the accepted box and coordinates, or the old ones back after a rejected trial
            first = nxt
    finally:
        if thermostat is not None:
            sys._check(L.mhip_set_andersen(sys._ctx, 0.0, 0.0, 0))
    if check_nans:
        sys._check(L.mhip_check_finite(sys._ctx))
    sys.pull_state()
    return sys


def random_velocities(sys, temp, rng=None):
    """random_velocities!(sys, temp; rng) (spatial.jl:803-831): Maxwell-Boltzmann velocities at temperature temp (K), drawn on the device"""
    L = _lib.lib()
    rng = _rng(rng)
    sys.push_state(velocities=True)
    ctr1, key = _rand_u64(rng), _rand_u64(rng)
    sys._check(L.mhip_random_velocities(sys._ctx, BOLTZMANN * float(temp), key, ctr1))
    sys.pull_state()
    return sys


def apply_coupling(sys, coupling, sim, rng=None, step_n=0):
    """apply_coupling!(sys, buffers, coupling, sim, neighbors, step_n; rng): one application.  AndersenThermostat (coupling.jl:196-211) returns False;
    MonteCarloBarostat (coupling.jl:861-1033) returns whether a move was accepted (the forces must be recomputed)."""
    if isinstance(coupling, MonteCarloBarostat):
        return _apply_mc_barostat(sys, coupling, step_n, _rng(rng))
    thermostat = coupling
    if not isinstance(thermostat, AndersenThermostat):
        raise MollyHipError(-6, f"coupling {type(thermostat).__name__} is outside the hot-path scope")
    L = _lib.lib()
    rng = _rng(rng)
    sys.push_state(velocities=True)
    ctr1, key = _rand_u64(rng), _rand_u64(rng)
    sys._check(L.mhip_andersen(sys._ctx, BOLTZMANN * float(thermostat.temperature), float(sim.dt) / float(thermostat.coupling_const), key, ctr1))
    sys.pull_state()
    return False


# ---- the boundary as a variable: what the barostats of coupling.jl need from the engine ---------------------------------------------------------------
BAR = 0.0602214076      # 1 bar in kJ mol⁻¹ nm⁻³ (1e5 Pa · 1e-27 m³/nm³ · N_A / 1e3)


def volume(boundary):
    """volume(boundary) (spatial.jl:365-372): the product of the side lengths; TriclinicBoundary: v1.x · v2.y · v3.z"""
    return float(np.prod(np.asarray(boundary.side_lengths, dtype=np.float64)))


def scale_boundary(boundary, scale):
    """scale_boundary(boundary, scale) (spatial.jl:414-422): a number or one factor per axis"""
    sc = np.broadcast_to(np.asarray(scale, dtype=np.float64), (3,))
    if isinstance(boundary, TriclinicBoundary):
        bv = boundary.basis_vectors
        return TriclinicBoundary(bv[0] * sc, bv[1] * sc, bv[2] * sc, approx_images=boundary.approx_images)
    return CubicBoundary(*(boundary.side_lengths * sc))


def scale_coords(sys, scale_matrix, scale_velocities=False):
    """scale_coords!(sys, μ; ignore_molecules=true) (spatial.jl:1184-1210, the branch of systems without a topology — every atom a molecule): box B′ = μ B,
    r′ = μ r in the system's number type, optionally v′ = μ⁻¹ v.  The rigid-molecule branch (:1211-1290) is host code on molecule lists and stays outside the
    engine; what the engine must do is follow the boundary (mhip_set_box)."""
    mu = np.asarray(scale_matrix, dtype=sys.dtype).reshape(3, 3)
    diagonal = not np.any(mu != np.diag(np.diag(mu)))
    b = sys.boundary
    if isinstance(b, TriclinicBoundary):
        nb = mu.astype(np.float64) @ b.basis_vectors.T            # columns = basis vectors (boxmatrix, spatial.jl:254)
        sys.boundary = TriclinicBoundary(nb[:, 0], nb[:, 1], nb[:, 2], approx_images=b.approx_images)
    else:
        if not diagonal:
            raise ValueError("a CubicBoundary is scaled by a diagonal matrix")
        sl = (np.diag(mu) * b.side_lengths.astype(sys.dtype)).astype(sys.dtype)   # μ · B in T, as the SMatrix product
        sys.boundary = CubicBoundary(*[float(x) for x in sl])
    sys.coords[:] = sys.coords * np.diag(mu)[None, :] if diagonal else (sys.coords @ mu.T).astype(sys.dtype)
    if scale_velocities:
        mi = np.linalg.inv(mu.astype(np.float64)).astype(sys.dtype)
        sys.velocities[:] = (sys.velocities @ mi.T).astype(sys.dtype)
    return sys


class MonteCarloBarostat:
    """MonteCarloBarostat(pressure, temperature, boundary; coupling_type=:isotropic, n_steps=30, n_iterations=1, scale_factor=0.01, scale_increment=1.1,
    max_volume_frac=0.3, trial_find_neighbors=false) (coupling.jl:721-859).  pressure in bar: a number (isotropic) or the three diagonal entries Pxx, Pyy, Pzz
    (semiisotropic, anisotropic).  The trial energies are full potential energies of the engine on the scaled box (the GPU path of the reference evaluates
    the cutoff sphere from the coordinates at every call, so `trial_find_neighbors` has nothing to choose there; it is kept for the signature)."""

    def __init__(self, pressure, temperature, boundary, coupling_type="isotropic", n_steps=30, n_iterations=1, scale_factor=0.01,
                 scale_increment=1.1, max_volume_frac=0.3, trial_find_neighbors=False):
        if coupling_type not in ("isotropic", "semiisotropic", "anisotropic"):      # coupling.jl:788-790
            raise ValueError("coupling_type must be :isotropic, :semiisotropic, or :anisotropic")
        p = np.asarray(pressure, dtype=np.float64)
        if coupling_type == "isotropic":
            if p.ndim != 0:
                raise ValueError("isotropic pressure must be a scalar")
            p = np.full(3, float(p))
        elif p.shape != (3,):
            raise ValueError(f"{coupling_type} pressure must have the three diagonal entries Pxx, Pyy, Pzz")
        self.pressure = p.copy()                                  # bar
        self.temperature = float(temperature)
        self.coupling_type = coupling_type
        self.n_steps, self.n_iterations = int(n_steps), int(n_iterations)
        self.volume_scale = volume(boundary) * float(scale_factor)
        self.scale_increment, self.max_volume_frac = float(scale_increment), float(max_volume_frac)
        self.trial_find_neighbors = bool(trial_find_neighbors)
        self.n_attempted = 0
        self.n_accepted = 0


def _mc_attempt(sys, barostat, E, rand, n_molecules):
    """one trial of apply_coupling_mc! (coupling.jl:886-1033) up to the trial state: scales sys, returns (dW-without-ΔE as a function of E_trial, undo)"""
    T = sys.dtype.type
    V = T(volume(sys.boundary))
    dV = T(barostat.volume_scale) * (T(2) * rand() - T(1))
    v_scale = (V + dV) / V
    kT = BOLTZMANN * barostat.temperature
    P = barostat.pressure * BAR
    if barostat.coupling_type == "isotropic":
        l = np.cbrt(v_scale)
        diag = (l, l, l)
        work = float(P.sum()) * float(dV) / 3.0                                   # tr(P) · dV / 3
    elif barostat.coupling_type == "semiisotropic":
        w1, w2 = rand(), rand()
        s_ = w1 + w2; w1, w2 = w1 / s_, w2 / s_
        lxy, lz = v_scale ** w1, v_scale ** w2
        diag = (lxy, lxy, lz)
        work = float((w1 / T(2)) * P[0] + (w1 / T(2)) * P[1] + w2 * P[2]) * float(V + dV) * math.log(float(v_scale))
    else:
        w1, w2, w3 = rand(), rand(), rand()
        s_ = w1 + w2 + w3; w1, w2, w3 = w1 / s_, w2 / s_, w3 / s_
        diag = (v_scale ** w1, v_scale ** w2, v_scale ** w3)
        work = float(w1 * P[0] + w2 * P[1] + w3 * P[2]) * float(V + dV) * math.log(float(v_scale))
    old_coords, old_boundary = sys.coords.copy(), sys.boundary
    scale_coords(sys, np.diag(np.asarray(diag, dtype=sys.dtype)))
    dW_of = lambda E_trial: (E_trial - E) + work - n_molecules * kT * math.log(float(v_scale))

    def undo():
        sys.coords[:] = old_coords
        sys.boundary = old_boundary
    return dW_of, undo, kT


def _apply_mc_barostat(sys, barostat, step_n, rng, energy=None):
    """apply_coupling!(sys, buffers, ::MonteCarloBarostat, sim, neighbors, step_n; rng) (coupling.jl:861-884).  `energy`: the potential-energy function (tests
    put the oracle's here to replay the same random numbers on the CPU)."""
    if step_n % barostat.n_steps != 0:
        return False
    energy = energy or (lambda s: potential_energy(s, step_n))
    T = sys.dtype.type
    rand = lambda: T(rng.random(dtype=sys.dtype))                 # rand(rng, T)
    n_molecules = len(sys)                                        # sys.topology === nothing
    recompute = False
    for _ in range(barostat.n_iterations):
        E = energy(sys)
        dW_of, undo, kT = _mc_attempt(sys, barostat, E, rand, n_molecules)
        dW = dW_of(energy(sys))
        if dW <= 0 or float(rand()) < math.exp(-dW / kT):
            recompute = True
            barostat.n_accepted += 1
        else:
            undo()
        barostat.n_attempted += 1
    if barostat.n_attempted >= 10:                                # coupling.jl:871-881
        V_now = volume(sys.boundary)
        if barostat.n_accepted < 0.25 * barostat.n_attempted:
            barostat.volume_scale /= barostat.scale_increment
        elif barostat.n_accepted > 0.75 * barostat.n_attempted:
            barostat.volume_scale = min(barostat.volume_scale * barostat.scale_increment, V_now * barostat.max_volume_frac)
        barostat.n_attempted = 0
        barostat.n_accepted = 0
    return recompute


def wrap_coords(coords, boundary):
    """wrap_coords(v, boundary) (spatial.jl:573-602)"""
    c = np.asarray(coords)
    if isinstance(boundary, TriclinicBoundary):
        T = c.dtype.type
        bv = boundary.basis_vectors.astype(c.dtype); rs = (T(1) / np.array([bv[0, 0], bv[1, 1], bv[2, 2]], dtype=c.dtype)).astype(c.dtype)
        a, b, cc = boundary.basis_vectors
        cot_bc = T(abs((b[1] * cc[1] + b[2] * cc[2]) / (b[1] * cc[2] - b[2] * cc[1])))
        cxz, cyz, cot_ab = T(cc[0] / abs(cc[2])), T(cc[1] / abs(cc[2])), T(b[0] / b[1])
        v = c.reshape(-1, 3).copy()
        v -= bv[2] * np.floor(v[:, 2] * rs[2])[:, None]
        v -= bv[1] * np.floor((v[:, 1] - v[:, 2] * cot_bc) * rs[1])[:, None]
        dx, dy = v[:, 2] * cxz, v[:, 2] * cyz
        v -= bv[0] * np.floor((v[:, 0] - dx - (v[:, 1] - dy) * cot_ab) * rs[0])[:, None]
        return v.reshape(c.shape)
    L = boundary.side_lengths.astype(c.dtype)
    return c - np.floor(c / L) * L

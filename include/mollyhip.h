/*
 * mollyhip.h — C ABI of libmollyhip.so, the MI355X-native nonbonded force engine that sits
 * behind Molly.jl's `System` / `pairwise_inters` / `simulate!` API.
 *
 * Every entry point below replaces one reference interface (paths relative to the Molly.jl
 * checkout, v0.23.3); INTEGRATION.md shows the Julia `ccall` stubs that bind them.
 *
 *   mhip_forces               ≙ Molly.pairwise_forces_loop_gpu!   ext/MollyCUDAExt.jl:845, src/kernels.jl:91,
 *                                                                  caller src/force.jl:1228
 *   mhip_specific_forces      ≙ Molly.specific_forces_gpu!        src/kernels.jl:142-205, caller src/force.jl:1231
 *   mhip_potential_energy     ≙ Molly.pairwise_pe_loop_gpu!       ext/MollyCUDAExt.jl:936, src/kernels.jl:393,
 *                                                                  caller src/energy.jl:427
 *   mhip_specific_potential_energy ≙ specific_pe_gpu!             src/kernels.jl:430-567, caller src/energy.jl:431
 *   mhip_general_forces / mhip_general_potential_energy ≙ AtomsCalculators.forces! / potential_energy of the PME general
 *                               interaction, src/interactions/ewald.jl:873-944, callers src/force.jl:792-795, src/energy.jl:239-245
 *   mhip_kinetic_energy       ≙ kinetic_energy                    src/energy.jl:56-89
 *   mhip_remove_cm            ≙ Molly.remove_CM_motion!           ext/MollyCUDAExt.jl:2373, src/spatial.jl:901-929
 *   mhip_vv_run               ≙ simulate!(sys, ::VelocityVerlet)  src/simulators.jl:547-668 (loop 589-666)
 *   mhip_export_neighbors     ≙ find_neighbors → NeighborList     src/neighbors.jl:390-423, 665-693; src/types.jl:611-654
 *   mhip_set_exceptions       ≙ GPUNeighborFinder excluded/special src/neighbors.jl:104-115, 171-195
 *   mhip_set_atoms            ≙ Atom{…} fields charge, σ, ϵ, mass, λ   src/types.jl:466-475
 *   mhip_set_state/get_state  ≙ sys.coords / sys.velocities       src/types.jl:798-800
 *
 * Conventions: plain C, no exceptions across the ABI. Every function returns an int32 status
 * (0 = ok, <0 = enum mhip_status); the message is available from mhip_last_error(ctx).
 * Units: nm, ps, g/mol, kJ/mol, elementary charge (Molly strips Unitful units before touching
 * buffers: src/force.jl:844-846).  Atom indices are 0-BASED here (the Julia shim subtracts 1).
 * Coordinate / velocity / force arrays are packed xyzxyz… of `float` (precision 32) or `double`
 * (precision 64) — bit-compatible with Vector{SVector{3,T}} and with fs_mat (3×N column-major,
 * src/force.jl:623) — so the caller passes pointers with zero copies.  `mem_kind` says whether a
 * pointer is host or device (gfx950 HBM) memory.  A context is NOT thread-safe: one host thread
 * drives it (as Julia does for one System).  The context owns all device memory it allocates;
 * the caller owns every pointer it passes and may free it when the call returns.
 *
 * Environment (read when a context is created; the defaults are the product, every value below is exercised by a test of tests/):
 *   MOLLYHIP_DEBUG=1 | 2 | 3       1: list-maintenance decisions (searches, prunes, skin changes) on stderr; 2: drain the stream before every launch and name it
 *                                  on stderr (the last name a dying process printed faulted); 3: both
 *   MOLLYHIP_XFER_TIMEOUT_MS=n     bound of every in-kernel wait for a peer rank (default 2000)
 *   MOLLYHIP_OUTER_MARGIN_PM=n     margin of the outer pair list in picometres (default 200; 0: one list of radius r_list)
 *   MOLLYHIP_INNER_SKIN_PM=n       skin of the inner pair list in picometres (default 100, never more than r_list − cutoff); MOLLYHIP_INNER_SKIN_FIXED=1 stops it growing
 *   MOLLYHIP_LDS_BUDGET_KB=n       LDS a force pass may use for its tile (default 160): smaller values make it walk the tile in segments
 *   MOLLYHIP_GROUP_SPLIT=0|4       the group-split pair pass of small systems off / forced (default: automatic); MOLLYHIP_ADOPT_OUTER=0: a pruning pass behind every search
 *   MOLLYHIP_FUSE_STEP=0           plain pair passes write forces and a separate launch integrates (default: the pass integrates in its epilogue)
 *   MOLLYHIP_FUSE_GATHER_VV=0      the same for a small system's last force launch
 *   MOLLYHIP_REUSE_RUN_FORCES=0    every run recomputes the forces of its first step (see mhip_vv_run)
 *   MOLLYHIP_PME_FFT=0|1           PME transforms through hipFFT never / always (default: meshes with more than 512 points on an axis)
 *   MOLLYHIP_DEVICE_REPLAN=0       mhip_domain_run returns to the host planner for every re-plan (see mhip_set_domain)
 *   MOLLYHIP_HALO_WAITER=0         ranks sharing ONE device (tests): no one-workgroup waiter in front of a fused ghosted step; the blocks' own bounded waits order it, as across devices
 *   MOLLYHIP_PRUNE_LATE=0          inside mhip_domain_run a pruning pass of a ghosted sub-domain drains the stream for its summary instead of reading it behind an event
 * Host mirror only (the Python files of molly.jl_amd): MOLLYHIP_GHOST_MARGIN_PM, MOLLYHIP_ENGINE_LOOP, MOLLYHIP_HALO_FUSED, MOLLYHIP_HOST_PRUNE, MOLLYHIP_DIST_BACKEND,
 * MOLLYHIP_FORCE_DEVICE, MOLLYHIP_FORCE_DOMAIN (bench.py), MOLLYHIP_LIB_AB (another build of the library, tools/force_ab.py).  Builds with -DMHIP_STAMPS=1 only:
 * MOLLYHIP_DBG_TIMES, MOLLYHIP_DBG_DUMP (time stamps inside the block kernels).  Every other switch of rounds 1-5 lost its measurement and was removed with
 * its code in round 6 (profiles/r06_removed_switches.md keeps the numbers).
 */
#ifndef MOLLYHIP_H
#define MOLLYHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mhip_ctx mhip_ctx;

enum mhip_status {
    MHIP_OK               =  0,
    MHIP_ERR_INVALID      = -1,   /* bad argument / inconsistent configuration            */
    MHIP_ERR_HIP          = -2,   /* a HIP runtime call failed (message has the hipError)  */
    MHIP_ERR_STATE        = -3,   /* call order violated (e.g. forces before set_state)    */
    MHIP_ERR_CAPACITY     = -4,   /* caller buffer too small (export_neighbors)            */
    MHIP_ERR_NO_DEVICE    = -5,   /* no gfx950 device visible: the product path never falls back to CPU */
    MHIP_ERR_UNSUPPORTED  = -6,   /* feature outside the hot-path scope                    */
    MHIP_ERR_NAN          = -7    /* NaN detected by mhip_check_finite                     */
};

/* src/cutoffs.jl */
enum mhip_cutoff_kind {
    MHIP_CUTOFF_NONE              = 0,   /* NoCutoff               cutoffs.jl:15,31      */
    MHIP_CUTOFF_DISTANCE          = 1,   /* DistanceCutoff         cutoffs.jl:72-79      */
    MHIP_CUTOFF_SHIFTED_POTENTIAL = 2,   /* ShiftedPotentialCutoff cutoffs.jl:100-113    */
    MHIP_CUTOFF_SHIFTED_FORCE     = 3,   /* ShiftedForceCutoff     cutoffs.jl:133-150    */
    MHIP_CUTOFF_CUBIC_SPLINE      = 4,   /* CubicSplineCutoff      cutoffs.jl:172-200    */
    MHIP_CUTOFF_POLYNOMIAL        = 5    /* PolynomialCutoff       cutoffs.jl:226-253    */
};

/* src/interactions/coulomb.jl */
enum mhip_coul_kind {
    MHIP_COUL_NONE           = 0,
    MHIP_COUL_PLAIN          = 1,   /* Coulomb               coulomb.jl:32-120      */
    MHIP_COUL_REACTION_FIELD = 2,   /* CoulombReactionField  coulomb.jl:698-814     */
    MHIP_COUL_EWALD_DIRECT   = 3    /* CoulombEwald          coulomb.jl:1320-1441   */
};

enum mhip_mem_kind { MHIP_MEM_HOST = 0, MHIP_MEM_DEVICE = 1 };

/* The `pairwise_inters` tuple: LennardJones(+cutoff) and one Coulomb flavour.  Plain data so the
 * Julia shim can fill it from the interaction structs (lennard_jones.jl:25-47, coulomb.jl). */
typedef struct mhip_interactions {
    int32_t lj_enabled;            /* 1 if a LennardJones is in pairwise_inters                   */
    int32_t lj_cutoff_kind;        /* enum mhip_cutoff_kind                                       */
    double  lj_rc;                 /* dist_cutoff                                                 */
    double  lj_ra;                 /* dist_activation (cubic spline / polynomial)                 */
    double  lj_weight_special;     /* weight_special, lennard_jones.jl:34 (0.5 for Amber 1-4)     */
    int32_t coul_kind;             /* enum mhip_coul_kind                                         */
    int32_t coul_cutoff_kind;      /* cutoff of the plain Coulomb (ignored for RF / Ewald)        */
    double  coul_rc;               /* dist_cutoff (all flavours)                                  */
    double  coul_ra;               /* dist_activation, plain Coulomb with spline/polynomial       */
    double  coul_ke;               /* coulomb_const, 138.93545764 kJ mol^-1 nm (coulomb.jl:16)    */
    double  coul_weight_special;   /* 0.8333… for Amber 1-4                                       */
    double  rf_dielectric;         /* solvent_dielectric; +inf = conducting boundary              */
    double  ewald_alpha;           /* α = sqrt(-log(2 tol))/rc, coulomb.jl:1332                   */
    int32_t ewald_approx_erfc;     /* approximate_erfc (A&S 7.1.26), coulomb.jl:1384-1393         */
    int32_t reserved;
} mhip_interactions;

typedef struct mhip_config {
    int32_t precision;             /* 32 | 64: T of System{D,AT,T}                                */
    int32_t device_id;             /* HIP device ordinal (one process per GPU)                    */
    int64_t n_atoms;               /* atom capacity of this context (owned + ghost)               */
    double  box[3];                /* CubicBoundary side lengths (spatial.jl:40); for a           */
                                   /* non-periodic axis: extent of the local domain               */
    double  origin[3];             /* lower corner, used on non-periodic axes only (else 0)       */
    int32_t periodic[3];           /* 1 = periodic axis. 0 = open axis of a spatial sub-domain    */
                                   /* whose ghost atoms the host supplies already shifted         */
    int32_t rebuild_every;         /* neighbour rebuild cadence in steps (DistanceNeighborFinder  */
                                   /* n_steps, neighbors.jl:385; default 10)                      */
    double  r_list;                /* neighbour radius = dist_cutoff + dist_buffer (setup.jl:565);*/
                                   /* +inf or <=0: every pair interacts (NoNeighborList)          */
    mhip_interactions inter;
} mhip_config;

typedef struct mhip_stats {
    int64_t n_atoms, n_owned, n_ghost;
    int64_t n_rebuilds;            /* neighbour-structure rebuilds so far                         */
    int64_t n_force_calls;
    int64_t n_pairs_full;          /* entries of the full (both-directions) list = 2·L            */
    int64_t n_list_slots;          /* padded slots actually streamed per force pass               */
    int64_t n_blocks;              /* i-blocks (workgroups of the force kernel)                   */
    int64_t tile_atoms_total;      /* Σ over blocks of LDS tile sizes                             */
    int32_t block_atoms;           /* i-atoms per workgroup                                       */
    int32_t j_split;               /* waves sharing one i-atom's list                             */
    int32_t minimg_mode;           /* 1 = in-loop exact minimum image (small boxes)               */
    int32_t max_tile_atoms;
    double  last_rebuild_ms;       /* host wall time of the last rebuild                          */
    int64_t lds_bytes;             /* dynamic LDS of the force kernel                             */
    int64_t algorithmic_bytes_step;/* N(R_p+22w)+4L, SURVEY §8(d)                                 */
    int64_t force_pass_bytes;      /* N(R_p+3w)+4L: algorithmic bytes of ONE force-kernel launch   */
    /* HIP-event timings, filled while profiling is on (mhip_set_profiling):                        */
    /* stage 0 pair-force kernel, 1 tile/list build kernel (outer search), 2 integrator kernels,    */
    /* 3 sort+permute, 4 force passes that prune the outer list, 5 bonded kernels, 6 PME reciprocal */
    double  prof_ms[8];
    int64_t prof_calls[8];
    int64_t n_outer_builds;        /* searches with the outer radius (dual pair list)             */
    int64_t n_filter_passes;
    int64_t tile_segments;         /* LDS segments the largest tile of the last force pass was walked in (1: resident as a whole) */
    int64_t n_group_split_passes;  /* plain force passes that ran as the group-split launch of small systems (csrc/forces_gs.hip) */
    int32_t group_split;           /* groups per block of that launch (0: not in use for this system)                             */
    int32_t n_adopted_outer_lists; /* rebuilds whose outer list became the inner list without a pruning pass (nothing to prune: inner radius = r_list) */
    int64_t n_fused_steps;         /* steps of mhip_vv_run without an integrator launch: the pair pass integrated in its epilogue (k_forces STEP, fp32 one-type
                                      fluids) or, with bonded terms and PME, the step's last force launch did (k_gather_collect_vv) */
    /* list upkeep of the dual pair list, priced like the force pass (DESIGN §4 "Roofline and algorithmic bytes"; 2 bytes per list slot, padding included) */
    int64_t n_outer_slots;         /* padded slots of the outer list: what an outer search writes and a pruning pass reads            */
    int64_t outer_tile_atoms_total;/* Σ over blocks of the outer list's tile sizes                                                    */
    int64_t build_pass_bytes;      /* ONE outer search: N·4w read + 2·n_outer_slots + 4·outer_tile_atoms_total written                */
    int64_t prune_pass_bytes;      /* ONE pruning force pass: N(R_p + 3w) + 2·n_outer_slots + 4·outer tile read, 2·n_list_slots + 4·tile_atoms_total + N·4w (snapshot) written */
    int64_t n_box_changes;         /* boundaries taken over by mhip_set_box (a barostat's trial moves, accepted or taken back)      */
} mhip_stats;

/* ---- lifetime ------------------------------------------------------------------------------ */
int32_t     mhip_create(mhip_ctx** out, const mhip_config* cfg);
int32_t     mhip_destroy(mhip_ctx* ctx);
const char* mhip_last_error(const mhip_ctx* ctx);            /* ctx may be NULL (create errors)  */
int32_t     mhip_device_count(int32_t* n_out);               /* visible HIP devices              */
int32_t     mhip_set_stream(mhip_ctx* ctx, void* hip_stream);/* run on the caller's hipStream_t  */
int32_t     mhip_synchronize(mhip_ctx* ctx);
/* per-stage hipEvent timers on the context's stream (≙ benchmark/gpu_profile_utils.jl:12-18); enabling
 * resets the accumulated numbers reported by mhip_get_stats */
int32_t     mhip_set_profiling(mhip_ctx* ctx, int32_t enable);

/* ---- topology / parameters ----------------------------------------------------------------- */
/* owned atoms come first, ghosts after them; n_owned + n_ghost <= cfg.n_atoms.  Default:
 * n_owned = cfg.n_atoms, n_ghost = 0. */
int32_t mhip_set_atom_counts(mhip_ctx* ctx, int64_t n_owned, int64_t n_ghost);
/* n_owned+n_ghost entries each, element type T; lambda may be NULL (λ = 1).  λ == 0 triggers the
 * LJZeroShortcut (mixing.jl:7-11). */
int32_t mhip_set_atoms(mhip_ctx* ctx, const void* charge, const void* sigma, const void* eps,
                       const void* mass, const void* lambda, int32_t mem_kind);
/* excluded (eligible == false) and special (1-4) pairs, i < j, host int32 arrays, as the
 * GPUNeighborFinder stores them (neighbors.jl:104-115). A pair in both lists is excluded. */
int32_t mhip_set_exceptions(mhip_ctx* ctx, const int32_t* ex_i, const int32_t* ex_j, int64_t n_ex,
                            const int32_t* sp_i, const int32_t* sp_j, int64_t n_sp);

/* specific_inter_lists (host arrays; parameters of element type T) */
int32_t mhip_set_bonds(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j,
                       const void* k, const void* r0);                 /* HarmonicBond  harmonic_bond.jl:44-54  */
int32_t mhip_set_angles(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j,
                        const int32_t* k, const void* kth, const void* th0); /* HarmonicAngle harmonic_angle.jl:46-67 */
/* one entry per (torsion, Fourier term); proper and improper torsions share this list */
int32_t mhip_set_torsions(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j,
                          const int32_t* k, const int32_t* l, const int32_t* periodicity,
                          const void* phase, const void* kt);          /* PeriodicTorsion periodic_torsion.jl:93-142 */
/* EwaldExclusion pairs (ewald.jl:978-1055); uses inter.ewald_alpha and inter.coul_ke */
int32_t mhip_set_ewald_exclusions(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j);

/* ---- state --------------------------------------------------------------------------------- */
/* xyz: 3·(n_owned+n_ghost) T; vel: 3·n_owned T or NULL (keep). Marks neighbour structures stale. */
int32_t mhip_set_state(mhip_ctx* ctx, const void* xyz, const void* vel, int32_t mem_kind);
int32_t mhip_get_state(mhip_ctx* ctx, void* xyz, void* vel, int32_t mem_kind);  /* either may be NULL */

/* ---- forces / energies --------------------------------------------------------------------- */
/* Pairwise LJ+Coulomb forces on the OWNED atoms at the current coordinates (3·n_owned T).
 * accumulate != 0 adds into f_xyz (the contract of pairwise_forces_loop_gpu!, whose caller
 * zeroes fs_mat, force.jl:1216); 0 overwrites.  step_n drives the rebuild cadence exactly like
 * find_neighbors (neighbors.jl:396): rebuild when stale or step_n % rebuild_every == 0 and the
 * step differs from the step of the last build.  virial9 (nullable): 9 HOST doubles, row-major 3x3, to which the pairwise
 * virial Σ dr ⊗ f over the pair list is ADDED (≙ needs_vir, force.jl:848-852, 877-880; buffers.virial_nounits). */
int32_t mhip_forces(mhip_ctx* ctx, int64_t step_n, int32_t accumulate, void* f_xyz,
                    void* virial9, int32_t mem_kind);
int32_t mhip_specific_forces(mhip_ctx* ctx, int32_t accumulate, void* f_xyz, int32_t mem_kind);
/* virial of the specific interactions, Σ r ⊗ f per term (≙ needs_vir in specific_force!, force.jl:991-1060), ADDED to 9 host doubles */
int32_t mhip_specific_virial(mhip_ctx* ctx, double* virial9);
int32_t mhip_potential_energy(mhip_ctx* ctx, int64_t step_n, double* pe_out);
int32_t mhip_specific_potential_energy(mhip_ctx* ctx, double* pe_out);
/* ---- general interaction: particle-mesh Ewald, reciprocal space ------------
 * ≙ PME(dist_cutoff, atoms, boundary; error_tol, order, ϵr) (ewald.jl:361-421).  The caller passes what that constructor
 * derives: B-spline `order` (4, 5 or 6; 0 switches PME off), mesh = pme_params.(box_sides, α, error_tol) (ewald.jl:479-482),
 * α = sqrt(-log(2 error_tol)) / dist_cutoff, ϵr; the Coulomb constant is inter.coul_ke.  Charges are the ones of mhip_set_atoms.
 * Once set, mhip_vv_run / vv_stage2 add the reciprocal-space forces after the specific interactions (force.jl:792-795).
 * Pairs with coul_kind = MHIP_COUL_EWALD_DIRECT (direct space) and mhip_set_ewald_exclusions (excluded-pair correction).
 * Single domain, fully periodic box only (MHIP_ERR_UNSUPPORTED otherwise). */
int32_t mhip_set_pme(mhip_ctx* ctx, int32_t order, const int32_t* mesh3, double alpha, double eps_r);
/* reciprocal-space forces, added to (accumulate != 0) or written into f_xyz   (ewald_pe_forces!, ewald.jl:873-916) */
int32_t mhip_general_forces(mhip_ctx* ctx, int32_t accumulate, void* f_xyz, int32_t mem_kind);
/* TriclinicBoundary(v1, v2, v3; approx_images) (spatial.jl:131-220; minimum image :528-551, wrap_coords :588-602): basis9 = the three
 * basis vectors row by row; v1 along x, v2 in the xy plane, v3.z > 0 (MHIP_ERR_INVALID otherwise, as the constructor's
 * ArgumentError).  The context must have been created with box = (v1.x, v2.y, v3.z), all axes periodic.  Call before mhip_set_state.
 * Single domain (MHIP_ERR_UNSUPPORTED otherwise); PME works on it (recip_box = invert_box_vectors, spatial.jl:338-347).  The neighbour search runs on a cell grid in fractional coordinates scaled
 * by the cell's perpendicular heights; a box too small for any grid (fewer than 6 cells of r_list / 2 on every axis) is handled as one
 * cell — every distance by the exact in-loop minimum image — and is then limited to 32 759 atoms.
 * approx_images != 0: the three-floor formula; 0: the search over the 27 neighbouring images. */
int32_t mhip_set_triclinic(mhip_ctx* ctx, const double* basis9, int32_t approx_images);
/* A new boundary for a live context: `sys.boundary = …` as scale_coords! does it for the barostats (spatial.jl:1184-1218 → coupling.jl:861-930; scale_boundary,
 * spatial.jl:414-422).  The reference's entry points read sys.boundary at every call (ext/MollyCUDAExt.jl:845, 936; force.jl:1196-1257), so a Monte-Carlo or
 * Berendsen barostat that stays in Julia works through this engine only if the engine follows the box: box3 = the new side lengths (a TriclinicBoundary:
 * v1.x, v2.y, v3.z), basis9 = the new basis vectors row by row for a context that has a TriclinicBoundary, NULL otherwise (MHIP_ERR_INVALID when the two
 * disagree; the image mode given to mhip_set_triclinic stays).  Kept: atoms and parameters, exception and bonded lists, the PME order / mesh / α (PME(...)
 * stores them, ewald.jl:285-309 — only recip_box follows the boundary), the launch shape, the velocities.  Made again: cell grid, capacities, reciprocal
 * box; every pair list is dropped.  The coordinates must be handed over again — they were scaled with the box — before the next force / energy / run call:
 * mhip_set_state (MHIP_ERR_STATE otherwise).  Single domain (MHIP_ERR_UNSUPPORTED for a context with ghosts or a domain plan: plan the bricks again).
 * A box the engine refuses leaves the context on its old box. */
int32_t mhip_set_box(mhip_ctx* ctx, const double* box3, const double* basis9);
/* E_recip + E_self + E_net-charge                                           (ewald.jl:898-928) */
int32_t mhip_general_potential_energy(mhip_ctx* ctx, double* pe_out);
/* reciprocal-space virial (recip_conv_inner!, ewald.jl:701-723, 747-750) + the net-charge term (:925-927), ADDED to 9 host doubles */
int32_t mhip_general_virial(mhip_ctx* ctx, double* virial9);
int32_t mhip_kinetic_energy(mhip_ctx* ctx, double* ke_out);
int32_t mhip_remove_cm(mhip_ctx* ctx);
int32_t mhip_check_finite(mhip_ctx* ctx);          /* check_nans, simulators.jl:98-111 */

/* ---- integrator ---------------------------------------------------------------------------- */
/* Runs steps first_step+1 … first_step+n_steps of velocity Verlet entirely on the device:
 *   v += a dt/2; x += v dt; x = wrap(x); F = forces(x); a' = F/m; v += a' dt/2;
 *   if remove_cm_every>0 && step % remove_cm_every == 0: v -= Σmv/Σm;  neighbour rebuild when
 *   step % rebuild_every == 0                                            (simulators.jl:589-666).
 * Like simulate!, the call first wraps the coordinates, removes CM motion when first_step == 0,
 * force-rebuilds the neighbour structures and recomputes the forces at first_step
 * (simulators.jl:561-571), so chunked calls continue a run (test/simulation.jl:16-57).
 * The work behind that contract is done only where its result could differ: lists that are provably still valid are kept (DESIGN §4,
 * "list lifetime at the boundary"), and a call that continues from exactly the state the previous mhip_vv_run left — no other
 * coordinates, no new parameters or terms, no search or prune due — takes that run's last forces for its first step (the same
 * numbers the recomputation yields: same lists, same order).  MOLLYHIP_REUSE_RUN_FORCES=0 always recomputes. */
int32_t mhip_vv_run(mhip_ctx* ctx, int64_t first_step, int64_t n_steps, double dt,
                    int32_t remove_cm_every);
/* The same step split at the point where a multi-GPU host exchanges ghost coordinates:
 *   stage1: v += a dt/2; x += v dt; x = wrap(x)           (owned atoms)
 *   stage2: F = forces(x) [+ specific]; v += a' dt/2      (owned atoms; ghosts read-only)     */
int32_t mhip_vv_init(mhip_ctx* ctx, int64_t first_step);          /* wrap, rebuild, forces at first_step */
int32_t mhip_vv_stage1(mhip_ctx* ctx, double dt);
int32_t mhip_vv_stage2(mhip_ctx* ctx, int64_t step_n, double dt);   /* incl. find_neighbors when step_n % rebuild_every == 0 */
int32_t mhip_rebuild(mhip_ctx* ctx, int64_t step_n);              /* force a neighbour rebuild now */

/* ---- stochastic dynamics ------------------------------------------------ */
/* All noise is Philox4x32-10 keyed by (key, ctr1) with the 1-based caller index of the atom as ctr0 (kernels.jl:688-741): results do
 * not depend on the internal atom order.  kT = k·temperature in kJ/mol.
 * simulate!(sys, ::Langevin, n_steps) (simulators.jl:1099-1220): per step
 *   F = forces(x); v += (F/m) dt; x += v dt/2; v = exp(−γ dt) v + sqrt(1 − exp(−2 γ dt)) sqrt(kT/m) ξ; x += v dt/2; x = wrap(x);
 *   CM removal and neighbour cadence as mhip_vv_run.  Step first_step + s (s = 1 …) uses ctr1 + s − 1 (:1189-1190). */
int32_t mhip_langevin_run(mhip_ctx* ctx, int64_t first_step, int64_t n_steps, double dt, double kT, double friction,
                          int32_t remove_cm_every, uint64_t philox_key, uint64_t philox_ctr1);
/* random_velocities!(sys, temp) (spatial.jl:803-831, random_velocities_kernel! kernels.jl:688-704): v_i = sqrt(kT/m_i) ξ_i */
int32_t mhip_random_velocities(mhip_ctx* ctx, double kT, uint64_t philox_key, uint64_t philox_ctr1);
/* One application of AndersenThermostat (coupling.jl:196-211, apply_andersen_coupling_kernel! kernels.jl:706-723): each atom is
 * re-drawn with probability prob = dt / coupling_const (clamped to [0, prevfloat(1)]). */
int32_t mhip_andersen(mhip_ctx* ctx, double kT, double prob, uint64_t philox_key, uint64_t philox_ctr1);
/* The same thermostat as the `coupling` of mhip_vv_run / mhip_langevin_run: applied after every step's CM removal (simulators.jl:630,
 * 1208); the per-step (ctr1, key) are words of philox(step, 0; seed).  prob <= 0 switches it off. */
int32_t mhip_set_andersen(mhip_ctx* ctx, double kT, double prob, uint64_t seed);
/* the raw generator on the device (known-answer tests): out4 = philox4x32_10(ctr4, key2) */
int32_t mhip_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4);

/* ---- neighbour list export (bit-exact check) ------------------------------------------------ */
/* Half list, each unordered pair once with i < j (0-based), special flag as neighbors.jl:411.
 * Pair SET equals the reference's for the same-precision arithmetic; order is unspecified.
 * capacity too small → MHIP_ERR_CAPACITY with *n_out = required entries. */
int32_t mhip_export_neighbors(mhip_ctx* ctx, int32_t* i, int32_t* j, uint8_t* special,
                              int64_t capacity, int64_t* n_out);
/* Current sorted order: perm[s] = caller index of the atom stored at sorted slot s
 * (≙ buffers.morton_seq, reorder round-trip test/gpu_optimizations.jl:220-250). */
int32_t mhip_export_order(mhip_ctx* ctx, int32_t* perm, int64_t capacity);
int32_t mhip_get_stats(mhip_ctx* ctx, mhip_stats* out);

/* ---- launch shape of the search and pair kernels (≙ CUDALaunchConfig, src/cuda_config.jl:1-62) --------------------------------
 * A workgroup handles `block_atoms` consecutive (Hilbert-sorted) atoms with `j_split` waves sharing every atom's list
 * (block_atoms * j_split lanes, at most 1024 in fp32 / 512 in fp64).  The automatic choice goes by the atom count.
 *   mhip_set_launch_config       ≙ set_cuda_launch_config! / reset_cuda_launch_config! (src/cuda_config.jl:17-47): block_atoms in
 *                                  {64, 128, 256}, j_split a power of two; (0, 0) = automatic again.  The lists are rebuilt.
 *   mhip_optimize_launch_config  ≙ optimize_cuda_launch_config! (ext/MollyCUDAExt.jl:594-642; src/cuda_config.jl:53-62): times
 *                                  n_passes plain force passes for each of a small candidate set on the context's own atoms
 *                                  (lists rebuilt per shape, untimed), installs the fastest and reports every trial
 *                                  (us_per_pass < 0: the shape's tile does not fit the LDS).  *n_trials = trials made (may exceed
 *                                  max_trials; only the first max_trials are written).  Single-domain contexts only. */
typedef struct mhip_launch_trial {
    int32_t block_atoms;
    int32_t j_split;
    float us_per_pass;
} mhip_launch_trial;
int32_t mhip_set_launch_config(mhip_ctx* ctx, int32_t block_atoms, int32_t j_split);
int32_t mhip_optimize_launch_config(mhip_ctx* ctx, int32_t n_passes, mhip_launch_trial* trials, int32_t max_trials, int32_t* n_trials);

/* ---- multi-GPU halo helpers (device pointers; run on the context's stream) ------------------ */
/* out[3k..3k+2] = coords[idx[k]] + shift[3k..]  (idx: caller indices of owned atoms; shift NULL ok) */
int32_t mhip_gather_coords(mhip_ctx* ctx, const int32_t* idx_dev, const void* shift_dev,
                           int64_t n, void* out_dev);
/* coords[first + k] = in[3k..3k+2], k < n  (used to refresh the ghost range each step) */
int32_t mhip_scatter_coords(mhip_ctx* ctx, int64_t first, int64_t n, const void* in_dev);
/* out4 = {Σ m vx, Σ m vy, Σ m vz, Σ m} over owned atoms, as double[4] on the HOST */
int32_t mhip_cm_momentum(mhip_ctx* ctx, double* out4);
int32_t mhip_shift_velocities(mhip_ctx* ctx, const double* dv3);  /* v -= dv3 on owned atoms */
/* The same without a host round trip: out4_dev (device double[4]) receives this domain's {Σ m v, Σ m}; after the
 * host all-reduced it over the domains (RCCL), mhip_remove_cm_dev subtracts P/M of the device-resident total. */
int32_t mhip_cm_momentum_dev(mhip_ctx* ctx, double* out4_dev);
/* total4_dev is read by the next vv_stage1 / vv_halo_begin (or any state read): keep it untouched until then. */
int32_t mhip_remove_cm_dev(mhip_ctx* ctx, const double* total4_dev);
/* Long-lived ghost plans (no reference equivalent).  A host that hands over a ghost shell of width
 * r_list + margin calls mhip_set_ghost_margin(margin) BEFORE mhip_set_atom_counts.  The sub-domain then searches an
 * outer list of radius r_list + margin once per plan and re-prunes it to the reference's r_list list at the
 * find_neighbors cadence (steps % rebuild_every == 0, inside vv_stage2 / vv_halo_end), exactly like the single-domain
 * engine.  The plan (ownership, ghost set, outer list) stays valid while no owned or ghost atom moved more than margin/2:
 * mhip_plan_disp2_dev writes max |x - x_plan|^2 as ONE float into device memory (MAX-all-reduce it over the ranks,
 * re-plan when 2 sqrt(d2) nears margin).  +inf means "re-plan at every rebuild step" (margin 0, or an interaction
 * without a cutoff <= r_list).  A force pass that finds the margin already exceeded fails with MHIP_ERR_STATE. */
int32_t mhip_set_ghost_margin(mhip_ctx* ctx, double margin);
/* out_dev: float[2] = { max |x - x_plan|^2 since the plan's outer search, max |x - x_prune|^2 since the inner list was last pruned }.
 * With a ghost margin the HOST schedules the prunes (all ranks at the same step): when 2 sqrt(out[1]) is about to use up the skin
 * r_list - max cutoff it either calls mhip_request_prune — the next force pass then re-prunes the outer list, which needs
 * 2 sqrt(out[0]) <= margin at that moment — or re-plans. */
int32_t mhip_plan_disp2_dev(mhip_ctx* ctx, float* out_dev);
int32_t mhip_request_prune(mhip_ctx* ctx);
/* One MD step of a ghosted sub-domain in two launches-batches around the ghost exchange:
 *   vv_halo_begin = vv_stage1 + gather_coords;   vv_halo_end = scatter_coords + vv_stage2 (+ this rank's
 *   {Σ m v, Σ m} into cm_out4_dev when non-NULL, to be all-reduced and handed back through mhip_remove_cm_dev). */
int32_t mhip_vv_halo_begin(mhip_ctx* ctx, double dt, const int32_t* idx_dev, const void* shift_dev,
                           int64_t n_send, void* send_dev);
int32_t mhip_vv_halo_end(mhip_ctx* ctx, int64_t step_n, double dt, int64_t first_ghost, int64_t n_ghost,
                         const void* recv_dev, double* cm_out4_dev);
/* Optional, between vv_halo_begin and vv_halo_end — while the ghost coordinates are on the wire: the pair forces of the atom
 * blocks whose neighbourhood holds no ghost atom.  *launched = 1 if that part of the force pass was issued (vv_halo_end then only
 * does the blocks that need ghosts), 0 on steps where the pass must run in one piece (a search or a prune is due). */
int32_t mhip_vv_halo_interior(mhip_ctx* ctx, int64_t step_n, int32_t* launched);
/* The same without the finalize launch: cm_parts_dev receives n_parts (<= 1024) per-block partials {Σ m vx, Σ m vy, Σ m vz, Σ m}
 * (double[4·n_parts], unused ones zero).  The host SUM-all-reduces the whole array over the ranks and hands it back through
 * mhip_remove_cm_parts_dev; the next first kick re-sums it in fixed order.  The array must stay untouched until then. */
int32_t mhip_vv_halo_end_parts(mhip_ctx* ctx, int64_t step_n, double dt, int64_t first_ghost, int64_t n_ghost,
                               const void* recv_dev, double* cm_parts_dev, int32_t n_parts);
int32_t mhip_remove_cm_parts_dev(mhip_ctx* ctx, const double* total_parts_dev, int32_t n_parts);

/* ---- the fused form of the ghosted step: ONE engine call per step after the ghost exchange --------------------------------------
 * (≙ simulators.jl:594-629 for one sub-domain: second kick of step n, remove_CM_motion!, first kick and drift of step n + 1 in one
 * integrator launch, as mhip_vv_run does on a single domain.)
 *
 * The buffers of a ghost plan are registered once (mhip_set_halo_plan; device memory, valid until the next call or the next
 * mhip_set_atom_counts, which drops the plan).  The send buffer holds n_send_rows rows of 3 reals: row k is the coordinate of local
 * atom send_idx[k] + send_shift[k], or — send_idx[k] < 0 — one of the cm_rows rows after each peer's atoms that carry this rank's
 * {Σ m vx, Σ m vy, Σ m vz, Σ m} of the step before (4 doubles = cm_rows·3 reals, cm_rows = 3 in fp32, 2 in fp64; send_cm_pos lists
 * the n_send_cm row numbers, peer by peer, row 0 .. cm_rows-1 each).  The receive buffer mirrors it: recv_dst[k] >= 0 is the ghost
 * slot (first_ghost + recv_dst[k]) that row k fills, recv_dst[k] = -1 - (p·cm_rows + r) marks row r of peer p's sums (p = 0 ..
 * n_cm_peers-1).  With every other rank a peer (1, 2, 4, 8 bricks) the centre-of-mass sum needs no all-reduce of its own: the next
 * integrator launch adds the peers' rows to its own sum and removes v_cm one launch late, together with v_cm·dt from the
 * coordinates that were drifted with it — the scheme of mhip_vv_run; forces are translation invariant.
 *
 *   mhip_vv_halo_start : first kick + drift + pack (after mhip_vv_init, after a step that stopped, after a re-plan)
 *   [ghost exchange of the send buffer into the receive buffer; mhip_vv_halo_interior meanwhile]
 *   mhip_vv_halo_mid   : unpack, pair forces, second kick of step_n, then — unless flags bit 1 — first kick + drift of the next step
 *                        and pack.  flags bit 0: remove_CM_motion! at this step.  flags bit 1 ("stop"): end behind the second kick
 *                        with coordinates and velocities of step_n in place (rebuild cadence, end of a run); the step's Σ m v then
 *                        goes to cm_parts_dev as n_parts per-block partials for an all-reduce + mhip_remove_cm_parts_dev, exactly
 *                        as mhip_vv_halo_end_parts leaves it. */
typedef struct {
    int64_t first_ghost, n_recv_rows;
    const void* recv; const int32_t* recv_dst;
    int32_t n_cm_peers, cm_rows;
    const int32_t* send_idx; const void* send_shift;
    int64_t n_send_rows; void* send;
    const int32_t* send_cm_pos; int32_t n_send_cm;
} mhip_halo_plan;
int32_t mhip_set_halo_plan(mhip_ctx* ctx, const mhip_halo_plan* plan);
int32_t mhip_vv_halo_start(mhip_ctx* ctx, double dt);
int32_t mhip_vv_halo_mid(mhip_ctx* ctx, int64_t step_n, double dt, int32_t flags, double* cm_parts_dev, int32_t n_parts);
/* The prune / re-plan decision of a ghost plan by the engine's own criteria (tight inner skin + drift bound, as on a single domain):
 * out3_dev = { max |x - x_plan|^2, max |x - x_prune|^2, max |v|^2 of the owned atoms } as device floats; MAX-all-reduce them, hand
 * the result (host memory) to mhip_plan_decide on EVERY rank.  *action: 0 = nothing, 1 = the next force pass re-prunes the outer
 * list (already arranged), 2 = re-plan (migrate, new ghost plan).  check_in (nullable): with action 0, *check_in = k > 0 asks for the
 * next check k steps from now instead of at the next cadence step — the list cannot be vouched for over a whole interval, but for k
 * steps (0: regular cadence; NULL: such lists are re-pruned now).  Supersedes mhip_plan_disp2_dev + mhip_request_prune, which keep
 * scheduling against the reference's skin r_list - cutoff. */
int32_t mhip_plan_state_dev(mhip_ctx* ctx, float* out3_dev);
int32_t mhip_plan_decide(mhip_ctx* ctx, int64_t step_n, const float* reduced3, int32_t* action, int32_t* check_in);


/* ---- the ghost exchange INSIDE the engine: peer stores over xGMI, the step loop in C++ (the reference has no
 * multi-device path, README.md:54) ---------------------------------------------------------------------------------------------------
 * One process per GPU.  Every rank owns a receive region in fine-grained device memory (two halves of rows_capacity rows of 3 reals,
 * used alternately by the parity of the exchange number, behind a small header of sequence words) and maps every peer's region
 * through its IPC handle.  After a step's coordinates are packed ONE kernel stores each peer's rows straight into that peer's region
 * and raises this rank's sequence word there; the receiving side's next step waits for its senders' words on the device (bounded:
 * 2 s, then MHIP_ERR_STATE at the end of the call instead of a hung GPU).  No host call and no collective per step.
 *
 *   mhip_halo_region     : allocate (or reuse) this rank's region; ipc_handle_out receives MHIP_IPC_HANDLE_BYTES bytes to hand to
 *                          the peers (any transport: the decomposition is set up by the host once)
 *   mhip_halo_open_peer  : map rank r's region from its handle (every rank that sends to or receives from this one; for the
 *                          collective validity check: every rank)
 *   mhip_halo_selftest  : collective check of the mapped regions before the first run (see below)
 *   mhip_set_halo_routes : per ghost plan, after mhip_set_halo_plan: the send buffer's consecutive segments → (peer, first row in the
 *                          peer's half); the receive buffer of the plan is the region half itself, peer by peer in the same order
 *   mhip_domain_run      : a run of ghosted velocity-Verlet steps in one call — mhip_vv_halo_start, then per step interior blocks →
 *                          wait → mhip_vv_halo_mid → peer stores, with the collective prune / re-plan decision (mhip_plan_state_dev →
 *                          MAX over the ranks → mhip_plan_decide) taken every rebuild interval from numbers that are read one step
 *                          after they were measured, so that nothing drains the stream.  Returns after n_steps (*reason = 0) or
 *                          behind the second kick of the step after which ownership and ghosts must be re-planned (*reason = 1);
 *                          *steps_done steps were taken.  If the last step taken removes the centre-of-mass motion, cm_parts_dev
 *                          holds its n_parts partials for the all-reduce (mhip_remove_cm_parts_dev).  counters3 (nullable) is
 *                          incremented by {checks, prunes arranged, re-plans asked}.  A sub-domain without peers (one rank) needs
 *                          no region and no routes. */
#define MHIP_IPC_HANDLE_BYTES 64
typedef struct {
    int32_t n_peers;
    const int32_t* peer_rank;      /* [n_peers] host: the ranks, in the order of the send / receive buffer segments */
    const int64_t* send_rows;      /* [n_peers] host: rows of the send buffer for each peer (consecutive segments, momentum rows included) */
    const int64_t* dst_row;        /* [n_peers] host: first row in the PEER's region half where this rank's segment lands */
    const int64_t* recv_rows;      /* [n_peers] host: rows received from each peer (consecutive in this rank's half, same order) */
} mhip_halo_routes;
int32_t mhip_halo_region(mhip_ctx* ctx, int64_t rows_capacity, int32_t world, int32_t rank, void* ipc_handle_out);
int32_t mhip_halo_open_peer(mhip_ctx* ctx, int32_t rank, const void* ipc_handle);
/* Collective (every rank, once every region is mapped): one round of stores into every rank's region and a bounded wait for theirs.
 * *ok = 0: a peer's store did not become visible here within 2 s — keep the ghost exchange on the host (torch.distributed) then. */
int32_t mhip_halo_selftest(mhip_ctx* ctx, int32_t* ok);
int32_t mhip_set_halo_routes(mhip_ctx* ctx, const mhip_halo_routes* routes);
int32_t mhip_domain_run(mhip_ctx* ctx, int64_t first_step, int64_t n_steps, double dt, int32_t remove_cm_every, double* cm_parts_dev, int32_t n_parts,
                        int64_t* steps_done, int32_t* reason, int64_t* counters3);


/* ---- the re-plan INSIDE the engine (SURVEY §8(e) "Migration: every neighbour rebuild ... counts first, then payload"; no reference
 * counterpart: README.md:54) --------------------------------------------------------------------------------------------------------
 * mhip_domain_run used to return (*reason = 1) whenever ownership and ghosts had to be redone, and the host migrated atoms, chose ghosts
 * and registered a new plan (mhip_set_atom_counts / mhip_set_atoms / mhip_set_state / mhip_set_halo_plan / mhip_set_halo_routes).  A
 * context that knows the decomposition does all of that on the device, in the middle of the step that found the plan stale: behind the
 * unpack of the step's ghost rows, in front of its force pass — which prunes the freshly searched outer list, as a single domain's pass
 * does behind an outer search — and the run goes on.  Leavers (position, velocity, parameters, global id) and the new ghosts (shifted
 * coordinates + parameters) travel as peer stores into a plan area behind each receive region's row halves, their counts as rows of a
 * count matrix in the region headers; every order is fixed (stayers in local order, arrivals by source rank then sender order, ghosts by
 * source rank, direction, sender order) — the layout molly.jl_amd/domain.py's host planner makes, so both planners give the same sums.
 *
 *   mhip_set_domain    : after the FIRST plan has been set up by the host (atoms, state, halo plan, routes).  grid = bricks per axis
 *                        (world = their product <= 64; rank r owns brick (r % gx, (r / gx) % gy, r / (gx gy))), box = the periodic box
 *                        that is cut, r_ghost = r_list + ghost margin.  Cut axes must be open in the context (mhip_config.periodic = 0),
 *                        uncut ones periodic.  global_ids_dev (nullable: 0 .. n_owned-1): the global index of every owned atom, local
 *                        order, device int64.  MHIP_ERR_UNSUPPORTED when some rank is not a neighbour of every other (more than two
 *                        bricks on an axis): such decompositions keep the host planner.  NULL geometry switches the device planner off.
 *   mhip_domain_info   : out8 = { owned atoms, ghost atoms, re-plans made by the engine, atoms that arrived in them, host microseconds spent
 *                        planning (launches + the one read-back), host microseconds in the searches behind the plans, 0, 0 }
 *   mhip_domain_export : what a host planner keeps per owned atom, local order, device memory (either may be NULL): global ids
 *                        (int64[n_owned]) and {q, sigma, eps, mass} (real[n_owned][4]).  With mhip_get_state this is the whole
 *                        sub-domain (gather of a final state, hand-over to a host re-plan).
 * A re-plan that does not fit (a context's atom capacity, a receive region) fails on EVERY rank with MHIP_ERR_CAPACITY — all of them
 * see the same count matrix — before anything is committed.  MOLLYHIP_DEVICE_REPLAN=0 keeps the host planner. */
typedef struct {
    int32_t grid[3];
    int32_t rank;
    double box[3];
    double r_ghost;
} mhip_domain_geometry;
int32_t mhip_set_domain(mhip_ctx* ctx, const mhip_domain_geometry* geometry, const int64_t* global_ids_dev);
int32_t mhip_domain_info(mhip_ctx* ctx, int64_t* out8);
int32_t mhip_domain_export(mhip_ctx* ctx, int64_t* global_ids_dev, void* params4_dev);

#ifdef __cplusplus
}
#endif
#endif /* MOLLYHIP_H */

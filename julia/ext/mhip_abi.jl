# mhip_abi.jl — what both Julia front ends of libmollyhip.so share (include()d by ext/MollyHIPExt.jl and by MollyHIP/src/MollyHIP.jl): the C structs of
# include/mollyhip.h (same field order, LP64), the context table, error handling, and the pairwise_inters → mhip_interactions mapping.
# Needs `using Molly, Unitful` in the including module.
const libmollyhip = get(ENV, "MOLLYHIP_LIB", joinpath(get(ENV, "MOLLYHIP_ROOT", pwd()), "molly.jl_amd", "libmollyhip.so"))

# ---- C structs of include/mollyhip.h (same field order, LP64) -------------------------------------------------------
struct MhipInteractions
    lj_enabled::Int32; lj_cutoff_kind::Int32; lj_rc::Float64; lj_ra::Float64; lj_weight_special::Float64
    coul_kind::Int32; coul_cutoff_kind::Int32; coul_rc::Float64; coul_ra::Float64; coul_ke::Float64
    coul_weight_special::Float64; rf_dielectric::Float64; ewald_alpha::Float64; ewald_approx_erfc::Int32; reserved::Int32
end
struct MhipConfig
    precision::Int32; device_id::Int32; n_atoms::Int64
    box::NTuple{3, Float64}; origin::NTuple{3, Float64}; periodic::NTuple{3, Int32}
    rebuild_every::Int32; r_list::Float64; inter::MhipInteractions
end

# ---- context table ---------------------------------------------------------------------------------------------------
mutable struct HipContext
    ptr::Ptr{Cvoid}
    exceptions_generation::UInt64          # nf.cache_generation the engine's exception lists were built from
    bonded_sent::Bool
    boundary::Any                          # the sys.boundary the engine's box was made from (follow_boundary!)
    atoms::Any                             # the sys.atoms array the engine's parameters were read from (=== : a replaced array is pushed again)
    inters::Any                            # the sys.pairwise_inters tuple mhip_create fixed the kinds and cutoffs from (a replaced tuple: a new context)
    function HipContext(ptr, boundary=nothing)
        c = new(ptr, typemax(UInt64), false, boundary, nothing, nothing)
        finalizer(c) do x                   # GC-driven release; release!(sys) is the eager form
            x.ptr == C_NULL || ccall((:mhip_destroy, libmollyhip), Int32, (Ptr{Cvoid},), x.ptr)
            x.ptr = C_NULL
        end
        return c
    end
end
const CONTEXTS = IdDict{Any, HipContext}()      # System object → its engine context (the interactions that walk the neighbour list)
const CONTEXTS_NOLIST = IdDict{Any, HipContext}()   # System object → the context of its use_neighbors = false interactions (NoNeighborList, force.jl:1219-1224): every pair, no exceptions
const CONTEXTS_LOCK = ReentrantLock()

last_error(ptr) = unsafe_string(ccall((:mhip_last_error, libmollyhip), Cstring, (Ptr{Cvoid},), ptr))
check(c::HipContext, rc) = rc == 0 ? nothing : error("libmollyhip: ", last_error(c.ptr))     # ≙ error(...) of ext:733-739

# `sys.boundary = …` on a live system (scale_coords!, spatial.jl:1202; a barostat's rejected move, coupling.jl:930): the reference's entry points read
# sys.boundary at every call (ext/MollyCUDAExt.jl:845, 936), so every look-up of a context compares it with the boundary the engine holds and hands a new one
# over (mhip_set_box: grid, capacities and reciprocal box made again, lists dropped; the caller's mhip_set_state that follows brings the scaled coordinates).
function follow_boundary!(c::HipContext, b)
    c.boundary === b && return c
    if c.boundary !== nothing
        typeof(b).name === typeof(c.boundary).name || error("libmollyhip: a live System keeps its kind of boundary")
        if hasproperty(b, :basis_vectors)                                              # TriclinicBoundary (spatial.jl:151-161)
            bv = Float64[ustrip(b.basis_vectors[r][k]) for r in 1:3 for k in 1:3]
            box = Float64[bv[1], bv[5], bv[9]]
            check(c, ccall((:mhip_set_box, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.ptr, box, bv))
        else                                                                           # CubicBoundary (spatial.jl:40)
            box = Float64[ustrip(x) for x in b.side_lengths]
            check(c, ccall((:mhip_set_box, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.ptr, box, Ptr{Float64}(C_NULL)))
        end
    end
    c.boundary = b
    return c
end

function release!(sys; table=nothing)
    lock(CONTEXTS_LOCK) do
        for t in (table === nothing ? (CONTEXTS, CONTEXTS_NOLIST) : (table,))
            c = pop!(t, sys, nothing)
            c === nothing || finalize(c)
        end
    end
    return nothing
end

# ---- pairwise_inters → mhip_interactions ----------------------------------------------------------------------------
cutoff_kind(::NoCutoff) = Int32(0); cutoff_kind(::DistanceCutoff) = Int32(1); cutoff_kind(::ShiftedPotentialCutoff) = Int32(2)
cutoff_kind(::ShiftedForceCutoff) = Int32(3); cutoff_kind(::CubicSplineCutoff) = Int32(4); cutoff_kind(::PolynomialCutoff) = Int32(5)
cutoff_rc(::NoCutoff) = 0.0;                 cutoff_rc(c) = Float64(ustrip(c.dist_cutoff))           # cutoffs.jl:72-253
cutoff_ra(c::Union{CubicSplineCutoff, PolynomialCutoff}) = Float64(ustrip(c.dist_activation)); cutoff_ra(c) = 0.0

function interactions(inters::Tuple)
    lj = (Int32(0), Int32(0), 0.0, 0.0, 1.0)
    coul = (Int32(0), Int32(0), 0.0, 0.0, 138.93545764, 1.0, 1.0, 0.0, Int32(1))     # kind, cutoff, rc, ra, ke, w14, ε_rf, α, approx
    n_lj = n_coul = 0
    for inter in inters
        if inter isa LennardJones                                                    # lennard_jones.jl:25-47
            n_lj += 1
            lj = (Int32(1), cutoff_kind(inter.cutoff), cutoff_rc(inter.cutoff), cutoff_ra(inter.cutoff), Float64(inter.weight_special))
        elseif inter isa Coulomb                                                     # coulomb.jl:32-69
            n_coul += 1
            coul = (Int32(1), cutoff_kind(inter.cutoff), cutoff_rc(inter.cutoff), cutoff_ra(inter.cutoff),
                    Float64(ustrip(inter.coulomb_const)), Float64(inter.weight_special), 1.0, 0.0, Int32(1))
        elseif inter isa CoulombReactionField                                        # coulomb.jl:698-746
            n_coul += 1
            coul = (Int32(2), Int32(0), Float64(ustrip(inter.dist_cutoff)), 0.0, Float64(ustrip(inter.coulomb_const)),
                    Float64(inter.weight_special), Float64(inter.solvent_dielectric), 0.0, Int32(1))
        elseif inter isa CoulombEwald                                                # coulomb.jl:1320-1382
            n_coul += 1
            coul = (Int32(3), Int32(0), Float64(ustrip(inter.dist_cutoff)), 0.0, Float64(ustrip(inter.coulomb_const)),
                    Float64(inter.weight_special), 1.0, Float64(ustrip(inter.α)), Int32(inter.approximate_erfc))
        else
            error("MollyHIPExt: pairwise interaction $(typeof(inter)) is outside the engine's scope (LennardJones, Coulomb, CoulombReactionField, CoulombEwald)")
        end
    end
    (n_lj <= 1 && n_coul <= 1) || error("MollyHIPExt: at most one LennardJones and one Coulomb-type interaction")
    return MhipInteractions(lj[1], lj[2], lj[3], lj[4], lj[5], coul[1], coul[2], coul[3], coul[4], coul[5], coul[6], coul[7], coul[8], coul[9], 0)
end

# one timed candidate of mhip_optimize_launch_config (mhip_launch_trial)
struct MhipLaunchTrial
    block_atoms::Int32
    j_split::Int32
    us_per_pass::Float32
end

# MollyHIP.jl — the same engine behind an UNMODIFIED Molly.jl: a plain package that depends on Molly and plugs into its documented extension points — a general
# interaction (docs/src/documentation.md:1015-1066) that computes the pairwise forces / energies in libmollyhip.so, and a custom neighbour finder
# (docs/src/developer.md:30-64) that hands Molly's own CPU loop the engine's list.  Works for Array systems (host pointers) and ROCArray systems (device pointers).
# NOT EXECUTED in the build image (no Julia there); held to the header and to the reference's names by tests/test_integration_md.py.
module MollyHIP

using Molly, StaticArrays, Unitful
import AtomsCalculators
import Molly: masses
include(joinpath(@__DIR__, "..", "..", "ext", "mhip_abi.jl"))      # MhipInteractions, MhipConfig, HipContext, CONTEXTS, check, last_error, interactions, libmollyhip

# ---- the force provider: a general interaction --------------------------------------------------------------------------
# System(...; pairwise_inters=(), neighbor_finder=NoNeighborFinder(), general_inters=(HIPNonbonded(...),)): every stock simulator,
# logger and coupler keeps working; the pairwise work of each step happens in libmollyhip.
struct HIPNonbonded{I <: Tuple, D}
    pairwise_inters::I                                   # LennardJones / Coulomb / CoulombReactionField / CoulombEwald, as Molly defines them
    dist_cutoff::D                                       # r_list = dist_cutoff + dist_buffer of setup.jl:565
    n_steps::Int                                         # the reference's find_neighbors cadence (neighbors.jl:385, 396)
    excluded_pairs::Vector{Tuple{Int32, Int32}}          # 1-based, i < j (bonded 1-2 / 1-3 exclusions, setup.jl:787-804)
    special_pairs::Vector{Tuple{Int32, Int32}}           # 1-4 pairs, scaled by the interactions' weight_special
end
HIPNonbonded(inters; dist_cutoff, n_steps=10, excluded_pairs=Tuple{Int32, Int32}[], special_pairs=Tuple{Int32, Int32}[]) =
    HIPNonbonded(inters, dist_cutoff, n_steps, excluded_pairs, special_pairs)

on_host(sys) = sys.coords isa Array
xyzptr(a::Array) = Ptr{Cvoid}(pointer(a))               # Vector{SVector{3, T}} (with or without units) = packed xyz of T
xyzptr(a) = Ptr{Cvoid}(UInt(pointer(a)))                 # ROCArray: device pointer

function context!(sys::System{3}, inter::HIPNonbonded)
    c = lock(() -> get(CONTEXTS, sys, nothing), CONTEXTS_LOCK)
    c === nothing || return follow_boundary!(c, sys.boundary)      # (a barostat replaced sys.boundary: mhip_set_box)
    T = Molly.float_type(sys)
    b = sys.boundary
    cfg = MhipConfig(T == Float32 ? Int32(32) : Int32(64), Int32(0), length(sys), Tuple(Float64.(ustrip.(b.side_lengths))), (0.0, 0.0, 0.0),
                     (Int32(1), Int32(1), Int32(1)), Int32(inter.n_steps), Float64(ustrip(inter.dist_cutoff)), interactions(inter.pairwise_inters))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mhip_create, libmollyhip), Int32, (Ref{Ptr{Cvoid}}, Ref{MhipConfig}), out, cfg)
    rc == 0 || error("libmollyhip: ", last_error(C_NULL))
    c = HipContext(out[], b)
    at = Array(sys.atoms)
    q = T[a.charge for a in at]; σ = T[ustrip(a.σ) for a in at]; ϵ = T[ustrip(a.ϵ) for a in at]; λ = T[a.λ for a in at]
    m = T.(ustrip.(Array(masses(sys))))
    check(c, ccall((:mhip_set_atoms, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Int32), c.ptr, q, σ, ϵ, m, λ, 0))
    ex_i = Int32[p[1] - 1 for p in inter.excluded_pairs]; ex_j = Int32[p[2] - 1 for p in inter.excluded_pairs]
    sp_i = Int32[p[1] - 1 for p in inter.special_pairs];  sp_j = Int32[p[2] - 1 for p in inter.special_pairs]
    check(c, ccall((:mhip_set_exceptions, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Int64, Ptr{Int32}, Ptr{Int32}, Int64),
                   c.ptr, ex_i, ex_j, length(ex_i), sp_i, sp_j, length(sp_i)))
    lock(() -> (CONTEXTS[sys] = c), CONTEXTS_LOCK)
    return c
end

# ≙ the general-interaction contract (documentation.md:1029-1043): ADD the forces to fs (same shape as the coordinates, force
# units), add the virial to buffers.virial when needs_vir (documentation.md:1052-1060).  mhip_forces(accumulate = 1) adds in place;
# the lists live in the context and follow step_n (cadence) and the displacement checks, as in §3.
function AtomsCalculators.forces!(fs, sys::System{3}, inter::HIPNonbonded; neighbors=nothing, step_n=0, n_threads=Threads.nthreads(),
                                  buffers=nothing, needs_vir=false, kwargs...)
    c = context!(sys, inter)
    mk = on_host(sys) ? Int32(0) : Int32(1)
    vir = zeros(Float64, 9)
    GC.@preserve fs begin
        check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, xyzptr(sys.coords), C_NULL, mk))
        check(c, ccall((:mhip_forces, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Ptr{Float64}, Int32),
                       c.ptr, Int64(step_n), Int32(1), xyzptr(fs), needs_vir ? pointer(vir) : Ptr{Float64}(C_NULL), mk))
    end
    if needs_vir && buffers !== nothing                                                    # row-major 3×3 → SMatrix (column-major)
        T = Molly.float_type(sys)
        buffers.virial .+= SMatrix{3, 3, T}(T.(permutedims(reshape(vir, 3, 3)))) .* sys.energy_units
    end
    return fs
end

function AtomsCalculators.potential_energy(sys::System{3}, inter::HIPNonbonded; neighbors=nothing, step_n=0, n_threads=Threads.nthreads(), kwargs...)
    c = context!(sys, inter)
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, xyzptr(sys.coords), C_NULL, on_host(sys) ? Int32(0) : Int32(1)))
    pe = Ref{Float64}(0.0)
    check(c, ccall((:mhip_potential_energy, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), c.ptr, Int64(step_n), pe))
    return Molly.float_type(sys)(pe[]) * sys.energy_units
end

# ---- the engine's list for stock CPU forces!: a custom neighbour finder ---------------------------------------------------
# find_neighbors contract (neighbors.jl:33-48, developer.md:41-64): return a NeighborList of (i, j, special), 1-based, i < j; reuse
# current_neighbors between refreshes.  mhip_export_neighbors hands out exactly the reference's list for the current coordinates
# (bit-identical pair set, tests/test_gpu_6mrr.py), so the stock pairwise_forces_loop! (force.jl:828-969) can walk it — which is
# also how the neighbour parity check is run from Julia.
struct HIPNeighborFinder{N <: HIPNonbonded}
    engine::N                                            # carries dist_cutoff, n_steps, the exception lists and the interaction constants
end

function Molly.find_neighbors(sys::System{3}, nf::HIPNeighborFinder, current_neighbors=nothing, step_n::Integer=0, force_recompute::Bool=false;
                              n_threads::Integer=Threads.nthreads())
    if !(force_recompute || step_n % nf.engine.n_steps == 0 || isnothing(current_neighbors))
        return current_neighbors
    end
    c = context!(sys, nf.engine)
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, xyzptr(sys.coords), C_NULL, on_host(sys) ? Int32(0) : Int32(1)))
    n = Ref{Int64}(0)
    check(c, ccall((:mhip_export_neighbors, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{UInt8}, Int64, Ref{Int64}),
                   c.ptr, C_NULL, C_NULL, C_NULL, Int64(0), n))                             # count only
    i = Vector{Int32}(undef, n[]); j = Vector{Int32}(undef, n[]); sp = Vector{UInt8}(undef, n[])
    check(c, ccall((:mhip_export_neighbors, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{UInt8}, Int64, Ref{Int64}),
                   c.ptr, i, j, sp, Int64(n[]), n))
    neighbors = isnothing(current_neighbors) ? NeighborList() : current_neighbors
    empty!(neighbors)
    for k in 1:n[]
        push!(neighbors, (i[k] + Int32(1), j[k] + Int32(1), sp[k] != 0x00))                 # 0-based C indices → Julia
    end
    return neighbors
end

end # module

# MollyHIPExt.jl — package extension of Molly.jl for AMDGPU.jl `ROCArray` systems: the four generics that MollyCUDAExt overrides for CuArray
# (ext/MollyCUDAExt.jl:73, 845, 936, 2373), bound to libmollyhip.so through the C ABI of include/mollyhip.h, plus device-resident steppers.
# Install (a fork of Molly.jl): copy this file and mhip_abi.jl into Molly's ext/, add the two stanzas of julia/Project.toml.fragment to Molly's
# Project.toml ([weakdeps] AMDGPU, [extensions] MollyHIPExt = "AMDGPU" — as MollyCUDAExt = "CUDA" is declared, Project.toml:41-54), set
# ENV["MOLLYHIP_ROOT"] to the checkout that holds molly.jl_amd/libmollyhip.so.  `using Molly, AMDGPU` then loads it.
# NOT EXECUTED in the build image (no Julia there): tests/test_integration_md.py holds every ccall to the header, every overriding method head to the
# reference's (tests/golden/reference_signatures.json) and every Molly name used here to the reference's sources; julia/test/runtests.jl is the test
# to run on a box with Julia + AMDGPU.jl.
module MollyHIPExt

using Molly, AMDGPU, StaticArrays, Unitful, Random
using GPUArrays: AbstractGPUArray                                                    # (a dependency of Molly itself, src/Molly.jl:21: the generic methods' array type, for `invoke`)
import Molly: pairwise_forces_loop_gpu!, pairwise_pe_loop_gpu!, remove_CM_motion!, uses_gpu_neighbor_finder, simulate!,
              random_velocities!, from_device, masses, ustrip_vec

include("mhip_abi.jl")      # the C structs, the context table, check / last_error / release!, interactions(): shared with MollyHIP.jl

uses_gpu_neighbor_finder(::Type{<:ROCArray}) = true                                  # ≙ ext/MollyCUDAExt.jl:73; consumer setup.jl:1938

# (struct MhipLaunchTrial: mhip_abi.jl)
# ---- launch shape: optimize_cuda_launch_config! / set_cuda_launch_config! (src/cuda_config.jl:17-62; ext/MollyCUDAExt.jl:594-642) ----

# Times the candidate workgroup shapes of the search / pair kernels on this system's atoms and keeps the fastest (the reference returns
# the winning force_block_y; here the winner is the (block_atoms, j_split) pair, stored in the context like sys.launch_config).
function Molly.optimize_cuda_launch_config!(sys::System{3, <:ROCArray, T}) where T
    sys.neighbor_finder isa GPUNeighborFinder || return nothing                      # ≙ ext:595
    c = context!(sys)
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), C_NULL, 1))
    trials = Vector{MhipLaunchTrial}(undef, 16); n = Ref{Int32}(0)
    check(c, ccall((:mhip_optimize_launch_config, libmollyhip), Int32, (Ptr{Cvoid}, Int32, Ptr{MhipLaunchTrial}, Int32, Ref{Int32}),
                   c.ptr, Int32(20), trials, Int32(16), n))
    timed = filter(t -> t.us_per_pass > 0, trials[1:min(n[], 16)])
    isempty(timed) && return nothing
    best = argmin(t -> t.us_per_pass, timed)
    return (Int(best.block_atoms), Int(best.j_split))
end

set_hip_launch_config!(sys::System{3, <:ROCArray}; block_atoms::Integer=0, j_split::Integer=0) =    # (0, 0) ≙ reset_cuda_launch_config!
    (c = context!(sys); check(c, ccall((:mhip_set_launch_config, libmollyhip), Int32, (Ptr{Cvoid}, Int32, Int32), c.ptr, Int32(block_atoms), Int32(j_split))))

# ---- create / look up the context of a System --------------------------------------------------------------------------
# Vector{SVector{3,T}} on the device is bit-compatible with packed xyz (include/mollyhip.h): zero-copy pointer, mem_kind = 1
devptr(a::ROCArray) = Ptr{Cvoid}(UInt(pointer(a)))

function push_exceptions!(c::HipContext, nf::GPUNeighborFinder)                       # neighbors.jl:104-115: 1-based, i < j, on the device
    c.exceptions_generation == nf.cache_generation && return
    ex_i, ex_j = Array(nf.excluded_i) .- Int32(1), Array(nf.excluded_j) .- Int32(1)
    sp_i, sp_j = Array(nf.special_i) .- Int32(1), Array(nf.special_j) .- Int32(1)
    check(c, ccall((:mhip_set_exceptions, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Int64, Ptr{Int32}, Ptr{Int32}, Int64),
                   c.ptr, ex_i, ex_j, length(ex_i), sp_i, sp_j, length(sp_i)))
    c.exceptions_generation = nf.cache_generation                                    # ≙ buffers.sparse_pair_generation, ext:779
    nf.initialized = true
end

function push_atoms!(c::HipContext, sys::System{3, <:ROCArray, T}) where T          # Atom fields, types.jl:466-475
    at = Array(sys.atoms)
    q = T[a.charge for a in at]; σ = T[ustrip(a.σ) for a in at]; ϵ = T[ustrip(a.ϵ) for a in at]; λ = T[a.λ for a in at]
    m = T.(ustrip.(Array(masses(sys))))
    check(c, ccall((:mhip_set_atoms, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Int32), c.ptr, q, σ, ϵ, m, λ, 0))
    c.atoms = sys.atoms
end

# The reference reads sys.atoms, sys.boundary, sys.pairwise_inters and the neighbour finder's exception caches at EVERY call (ext/MollyCUDAExt.jl:845-873); the
# engine keeps them in its context, so every look-up checks that what it keeps is still what the System holds: a replaced boundary goes through mhip_set_box, a
# replaced atoms array through mhip_set_atoms, new exception pairs through mhip_set_exceptions, a replaced interaction tuple makes a new context.
function context!(sys::System{3, <:ROCArray, T}, inters::Tuple=sys.pairwise_inters; no_list::Bool=false) where T
    table = no_list ? CONTEXTS_NOLIST : CONTEXTS
    c = lock(CONTEXTS_LOCK) do
        get(table, sys, nothing)
    end
    nf = sys.neighbor_finder
    no_list || nf isa GPUNeighborFinder || error("MollyHIPExt expects the GPUNeighborFinder that setup picks for ROCArray systems (setup.jl:1938-1949)")
    if c !== nothing && c.inters !== inters
        release!(sys; table=table); c = nothing
    end
    if c === nothing
        b = sys.boundary
        tric = b isa TriclinicBoundary
        sides = tric ? (ustrip(b.basis_vectors[1][1]), ustrip(b.basis_vectors[2][2]), ustrip(b.basis_vectors[3][3])) :
                       Tuple(Float64.(ustrip.(b.side_lengths)))                      # spatial.jl:40, 151-161
        cfg = MhipConfig(T == Float32 ? Int32(32) : Int32(64), Int32(AMDGPU.device_id(AMDGPU.device()) - 1), length(sys),
                         Float64.(sides), (0.0, 0.0, 0.0), (Int32(1), Int32(1), Int32(1)),
                         no_list ? Int32(10) : Int32(nf.n_steps_reorder), no_list ? Inf : Float64(ustrip(nf.dist_cutoff)), interactions(inters))   # r_list = +Inf: every pair interacts (include/mollyhip.h)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mhip_create, libmollyhip), Int32, (Ref{Ptr{Cvoid}}, Ref{MhipConfig}), out, cfg)
        rc == 0 || error("libmollyhip: ", last_error(C_NULL))
        c = HipContext(out[], b)
        c.inters = inters
        check(c, ccall((:mhip_set_stream, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), c.ptr, AMDGPU.stream().stream))   # kernels join the task's HIP stream
        if tric
            bv = Float64[ustrip(b.basis_vectors[r][k]) for r in 1:3 for k in 1:3]
            approx = typeof(b).parameters[end] === true                              # TriclinicBoundary{D, T, A, …}: A = approx_images
            check(c, ccall((:mhip_set_triclinic, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32), c.ptr, bv, approx ? 1 : 0))
        end
        lock(CONTEXTS_LOCK) do
            table[sys] = c
        end
    end
    c.atoms === sys.atoms || push_atoms!(c, sys)                                     # first use, or `sys.atoms = …` since (mhip_set_atoms drops what depended on them)
    follow_boundary!(c, sys.boundary)                                                # a barostat replaced sys.boundary (coupling.jl:861-1033): mhip_set_box
    no_list || push_exceptions!(c, nf)                                               # also after append_excluded_pairs! (neighbors.jl:313); NoNeighborList carries no exceptions (kernels.jl:98-100)
    return c
end

# ---- the four overridden generics ------------------------------------------------------------------------------------
# ≙ ext/MollyCUDAExt.jl:845 — ACCUMULATES unit-less forces into buffers.fs_mat (3×N, zeroed by the caller, force.jl:1216) and
# the pair virial into buffers.virial_nounits (3×3 device matrix) when needs_vir (force.jl:1228, 1241).
function pairwise_forces_loop_gpu!(buffers, sys::System{3, <:ROCArray, T}, pairwise_inters::Tuple, nbs::Nothing,
                                   ::Val{needs_vir}, step_n) where {T, needs_vir}
    return engine_forces!(buffers, sys, context!(sys, pairwise_inters), Val(needs_vir), step_n)       # (the tuple handed in: forces! passes the use_neighbors = true subset, force.jl:1226-1229)
end

# ≙ ext/MollyCUDAExt.jl:757 (generic kernels.jl:91-100): the use_neighbors = false interactions over EVERY pair, as forces! hands them over with a NoNeighborList
# (force.jl:1219-1224).  The engine walks all pairs of one tile for up to 32 000 atoms (r_list = +Inf); beyond that the generic KernelAbstractions method keeps the call.
const NOLIST_MAX_ATOMS = 32_000
function pairwise_forces_loop_gpu!(buffers, sys::System{3, <:ROCArray, T}, pairwise_inters::Tuple, nbs::Molly.NoNeighborList,
                                   ::Val{needs_vir}, step_n) where {T, needs_vir}
    length(sys) <= NOLIST_MAX_ATOMS ||
        return invoke(pairwise_forces_loop_gpu!, Tuple{Any, System{3, <:AbstractGPUArray}, Any, Any, Val{needs_vir}, Any}, buffers, sys, pairwise_inters, nbs, Val(needs_vir), step_n)
    return engine_forces!(buffers, sys, context!(sys, pairwise_inters; no_list=true), Val(needs_vir), step_n)
end

function engine_forces!(buffers, sys::System{3, <:ROCArray, T}, c::HipContext, ::Val{needs_vir}, step_n) where {T, needs_vir}
    # coordinates may have changed since the last call (ext:778 "needs_reorder = true"); the engine keeps its lists if it can
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), C_NULL, 1))
    vir = zeros(Float64, 9)                                                          # row-major 3×3, host; mhip_forces ADDS Σ dr ⊗ f
    check(c, ccall((:mhip_forces, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Ptr{Float64}, Int32),
                   c.ptr, Int64(step_n), 1, Ptr{Cvoid}(UInt(pointer(buffers.fs_mat))), needs_vir ? pointer(vir) : C_NULL, 1))
    if needs_vir
        buffers.virial_nounits .+= ROCArray(T.(permutedims(reshape(vir, 3, 3))))    # symmetric; permutedims makes the layout explicit
    end
    buffers.step_n_preprocessed = step_n                                             # keep the reference's bookkeeping coherent (force.jl:517)
    return buffers
end

# ≙ ext/MollyCUDAExt.jl:936 — accumulates into the 1-element device vector (zeroed by the caller, energy.jl:417)
function pairwise_pe_loop_gpu!(pe_vec_nounits, buffers, sys::System{3, <:ROCArray, T}, pairwise_inters::Tuple, nbs::Nothing, step_n) where T
    return engine_energy!(pe_vec_nounits, sys, context!(sys, pairwise_inters), step_n)
end
# (energy.jl:419-423: the use_neighbors = false interactions with a NoNeighborList)
function pairwise_pe_loop_gpu!(pe_vec_nounits, buffers, sys::System{3, <:ROCArray, T}, pairwise_inters::Tuple, nbs::Molly.NoNeighborList, step_n) where T
    length(sys) <= NOLIST_MAX_ATOMS ||
        return invoke(pairwise_pe_loop_gpu!, Tuple{Any, Any, System{3, <:AbstractGPUArray}, Any, Any, Any}, pe_vec_nounits, buffers, sys, pairwise_inters, nbs, step_n)
    return engine_energy!(pe_vec_nounits, sys, context!(sys, pairwise_inters; no_list=true), step_n)
end

function engine_energy!(pe_vec_nounits, sys::System{3, <:ROCArray, T}, c::HipContext, step_n) where T
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), C_NULL, 1))
    pe = Ref{Float64}(0.0)
    check(c, ccall((:mhip_potential_energy, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Ref{Float64}), c.ptr, Int64(step_n), pe))
    pe_vec_nounits .+= T(pe[])
    return pe_vec_nounits
end

# ≙ ext/MollyCUDAExt.jl:2373 (generic spatial.jl:920): v .-= Σ m v / Σ m in place on sys.velocities.  Virtual sites (the
# `has_vs` branch of ext:2442-2449) are outside the engine's scope: such systems keep the generic method.
function remove_CM_motion!(sys::System{3, <:ROCArray, T}) where T
    isempty(sys.virtual_sites) || return invoke(remove_CM_motion!, Tuple{System}, sys)
    c = context!(sys)
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), devptr(sys.velocities), 1))
    check(c, ccall((:mhip_remove_cm, libmollyhip), Int32, (Ptr{Cvoid},), c.ptr))
    check(c, ccall((:mhip_get_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, C_NULL, devptr(sys.velocities), 1))
    return sys
end

# ---- device-resident steppers (custom simulators, docs/src/documentation.md:1202-1252) -----------------------------------------
# Bonded lists and the PME general interaction move into the engine once per context; state crosses the boundary only when a
# logger is due: the run is cut into chunks of the gcd of the logger intervals (apply_loggers! fires at its cadence, simulators.jl:657).
function push_specific!(c::HipContext, sys::System{3, <:ROCArray, T}) where T
    c.bonded_sent && return
    for sil in values(sys.specific_inter_lists)
        inters = Array(sil.inters); n = length(inters)
        i0(v) = Array(v) .- Int32(1)
        if sil isa InteractionList2Atoms && eltype(inters) <: HarmonicBond           # types.jl:89-101, harmonic_bond.jl:13-16
            check(c, ccall((:mhip_set_bonds, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{T}, Ptr{T}),
                           c.ptr, n, i0(sil.is), i0(sil.js), T[ustrip(b.k) for b in inters], T[ustrip(b.r0) for b in inters]))
        elseif sil isa InteractionList2Atoms && eltype(inters) <: EwaldExclusion     # ewald.jl:978; α, ϵr come from the context's CoulombEwald
            check(c, ccall((:mhip_set_ewald_exclusions, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}), c.ptr, n, i0(sil.is), i0(sil.js)))
        elseif sil isa InteractionList3Atoms && eltype(inters) <: HarmonicAngle      # harmonic_angle.jl:15-18
            check(c, ccall((:mhip_set_angles, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{T}, Ptr{T}),
                           c.ptr, n, i0(sil.is), i0(sil.js), i0(sil.ks), T[ustrip(a.k) for a in inters], T[ustrip(a.θ0) for a in inters]))
        elseif sil isa InteractionList4Atoms && eltype(inters) <: PeriodicTorsion    # periodic_torsion.jl:17-22: one engine term per Fourier term
            ti = Int32[]; tj = Int32[]; tk = Int32[]; tl = Int32[]; per = Int32[]; ph = T[]; k0 = T[]
            is, js, ks, ls = i0(sil.is), i0(sil.js), i0(sil.ks), i0(sil.ls)
            for (t, tor) in enumerate(inters), f in eachindex(tor.periodicities)
                iszero(ustrip(tor.ks[f])) && continue
                push!(ti, is[t]); push!(tj, js[t]); push!(tk, ks[t]); push!(tl, ls[t])
                push!(per, Int32(tor.periodicities[f])); push!(ph, T(tor.phases[f])); push!(k0, T(ustrip(tor.ks[f])))
            end
            check(c, ccall((:mhip_set_torsions, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{T}, Ptr{T}),
                           c.ptr, length(ti), ti, tj, tk, tl, per, ph, k0))
        else
            error("MollyHIPExt: specific interaction list $(typeof(sil)) is outside the engine's scope")
        end
    end
    for gi in values(sys.general_inters)
        gi isa PME || error("MollyHIPExt: general interaction $(typeof(gi)) is outside the engine's scope")
        mesh = Int32[gi.mesh_dims...]                                                # ewald.jl:285-309
        check(c, ccall((:mhip_set_pme, libmollyhip), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}, Float64, Float64),
                       c.ptr, Int32(gi.order), mesh, Float64(ustrip(gi.α)), Float64(gi.ϵr)))
    end
    c.bonded_sent = true
end

# The run is cut where something on the host side is due: a logger (its n_steps; apply_loggers! fires at the multiples, simulators.jl:657) or a coupling the engine does not
# carry — every coupling of coupling.jl except the AndersenThermostat, which is the engine's (mhip_set_andersen).  Those are applied by the reference's own
# apply_coupling! between two chunks (coordinates, velocities and the boundary go back to the engine behind it; context! follows a replaced boundary: mhip_set_box), at
# the multiples of their n_steps where they have one (MonteCarloBarostat) and after every step where they have not (the rescaling thermostats); couplings that read the
# virial of their step's force call are refused below.
couplers_of(sim) = sim.coupling === nothing ? () : (sim.coupling isa Union{Tuple, NamedTuple} ? Tuple(values(sim.coupling)) : (sim.coupling,))
host_couplers(sim) = Tuple(c for c in couplers_of(sim) if !(c isa AndersenThermostat))
host_intervals(sys, sim) = (Int[l.n_steps for l in values(sys.loggers) if hasproperty(l, :n_steps)]..., Int[hasproperty(c, :n_steps) ? c.n_steps : 1 for c in host_couplers(sim)]...)
next_stop(first, last, intervals) = minimum((last, ((fld(first, k) + 1) * k for k in intervals if k > 0)...))

function run_chunks!(step!, c::HipContext, sys::System{3, <:ROCArray, T}, sim, n_steps, init_step, run_loggers, rng) where T
    push_specific!(c, sys)
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), devptr(sys.velocities), 1))
    Molly.apply_loggers!(sys, nothing, init_step, nothing, run_loggers)              # loggers.jl:44, simulators.jl:572
    intervals = host_intervals(sys, sim)
    on_host = host_couplers(sim)
    # (a coupling that consumes the virial — the Berendsen and C-rescale barostats, needs_virial(c) = c.n_steps — reads it from the buffers of the force call of its step
    # (pressure(…; recompute=false), coupling.jl:403): the stock simulators make that call, these steppers keep the forces on the device — such couplings run under the
    # stock `VelocityVerlet` / `Langevin` loops, whose force and energy calls are the overrides above)
    for cpl in on_host
        isfinite(Molly.needs_virial(cpl)) && error("MollyHIPExt: $(typeof(cpl)) reads the virial of its step's force call; run it under Molly's own simulator loop (the ROCArray overrides serve it there)")
    end
    buffers = isempty(on_host) ? nothing : Molly.init_buffers!(sys, Threads.nthreads())
    first, last = init_step, init_step + n_steps
    while first < last
        stop = next_stop(first, last, intervals)
        step!(first, stop - first)                                                   # mhip_vv_run / mhip_langevin_run: returns after a stream sync
        check(c, ccall((:mhip_get_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), devptr(sys.velocities), 1))
        first = stop
        if !isempty(on_host)                                                         # ≙ simulators.jl:630, 1208: apply_coupling! at the end of step `first`
            Molly.apply_coupling!(sys, buffers, on_host, sim, nothing, first; n_threads=Threads.nthreads(), rng=rng)
            c = context!(sys)                                                        # (a barostat replaced sys.boundary: the engine follows before it sees the scaled coordinates)
            check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), devptr(sys.velocities), 1))
        end
        Molly.apply_loggers!(sys, nothing, first, nothing, run_loggers)              # loggers call forces / potential_energy → the overrides above
    end
    return sys
end

struct HIPVelocityVerlet{T, C}; dt::T; coupling::C; remove_CM_motion::Int; end
HIPVelocityVerlet(; dt, coupling=nothing, remove_CM_motion=1) = HIPVelocityVerlet(dt, coupling, Int(remove_CM_motion))   # ≙ VelocityVerlet, simulators.jl:280-300

function with_andersen(f, c::HipContext, sys, sim, rng)                              # coupling.jl:188-211: kT, P = dt/τ, per-step Philox words
    ths = Tuple(x for x in couplers_of(sim) if x isa AndersenThermostat)
    isempty(ths) && return f()
    length(ths) == 1 || error("MollyHIPExt: one AndersenThermostat per simulator")
    th = ths[1]
    check(c, ccall((:mhip_set_andersen, libmollyhip), Int32, (Ptr{Cvoid}, Float64, Float64, UInt64), c.ptr,
                   Float64(ustrip(th.temperature * sys.k)), Float64(ustrip(sim.dt / th.coupling_const)), rand(rng, UInt64)))
    try
        return f()
    finally
        ccall((:mhip_set_andersen, libmollyhip), Int32, (Ptr{Cvoid}, Float64, Float64, UInt64), c.ptr, 0.0, 0.0, UInt64(0))
    end
end

function simulate!(sys::System{3, <:ROCArray, T}, sim::HIPVelocityVerlet, n_steps::Integer;
                   init_step::Integer=0, run_loggers=true, rng=Random.default_rng(), kwargs...) where T
    c = context!(sys)
    with_andersen(c, sys, sim, rng) do
        run_chunks!(c, sys, sim, n_steps, init_step, run_loggers, rng) do first, n   # ≙ simulators.jl:589-666
            check(c, ccall((:mhip_vv_run, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Int64, Float64, Int32),
                           c.ptr, Int64(first), Int64(n), Float64(ustrip(sim.dt)), Int32(sim.remove_CM_motion)))
        end
    end
end

struct HIPLangevin{T, K, F, C}; dt::T; temperature::K; friction::F; coupling::C; remove_CM_motion::Int; end
HIPLangevin(; dt, temperature, friction, coupling=nothing, remove_CM_motion=1) = HIPLangevin(dt, temperature, friction, coupling, Int(remove_CM_motion))

function simulate!(sys::System{3, <:ROCArray, T}, sim::HIPLangevin, n_steps::Integer;
                   init_step::Integer=0, run_loggers=true, rng=Random.default_rng(), kwargs...) where T
    c = context!(sys)
    key, ctr1 = rand(rng, UInt64), rand(rng, UInt64)                                 # as simulators.jl:1149-1150 draws them
    kT = Float64(ustrip(sim.temperature * sys.k))
    with_andersen(c, sys, sim, rng) do
        run_chunks!(c, sys, sim, n_steps, init_step, run_loggers, rng) do first, n   # ≙ simulators.jl:1099-1220; ctr1 advances one per step (:1190)
            check(c, ccall((:mhip_langevin_run, libmollyhip), Int32, (Ptr{Cvoid}, Int64, Int64, Float64, Float64, Float64, Int32, UInt64, UInt64),
                           c.ptr, Int64(first), Int64(n), Float64(ustrip(sim.dt)), kT, Float64(ustrip(sim.friction)), Int32(sim.remove_CM_motion),
                           key, ctr1 + UInt64(first - init_step)))
        end
    end
end

# random_velocities!(sys, temp; rng) (spatial.jl:803-831) on the device
function random_velocities!(sys::System{3, <:ROCArray, T}, temp; rng=Random.default_rng()) where T
    c = context!(sys)
    ctr1, key = rand(rng, UInt64), rand(rng, UInt64)
    check(c, ccall((:mhip_set_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, devptr(sys.coords), devptr(sys.velocities), 1))
    check(c, ccall((:mhip_random_velocities, libmollyhip), Int32, (Ptr{Cvoid}, Float64, UInt64, UInt64), c.ptr, Float64(ustrip(temp * sys.k)), key, ctr1))
    check(c, ccall((:mhip_get_state, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.ptr, C_NULL, devptr(sys.velocities), 1))
    return sys
end

# specific + general virial for virial(sys) / pressure(sys) (energy.jl:116-131; spatial.jl:930-982): both ADD to 9 host doubles
function add_specific_and_general_virial!(buffers, sys::System{3, <:ROCArray, T}) where T
    c = context!(sys); push_specific!(c, sys)
    vir = zeros(Float64, 9)
    check(c, ccall((:mhip_specific_virial, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), c.ptr, vir))     # ≙ force.jl:991-1060
    check(c, ccall((:mhip_general_virial, libmollyhip), Int32, (Ptr{Cvoid}, Ptr{Float64}), c.ptr, vir))      # ≙ ewald.jl:701-723, 747-750, 925-927
    buffers.virial_nounits .+= ROCArray(T.(permutedims(reshape(vir, 3, 3))))
    return buffers
end

end # module

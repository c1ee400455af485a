# runtests.jl — the reference's own GPU-consistency checks, restated for `ROCArray` systems behind MollyHIPExt (julia/ext) and for the plugin route of
# MollyHIP (julia/MollyHIP).  NOT EXECUTED in the build image (no Julia there).  On a box with Julia >= 1.10, AMDGPU.jl, a gfx950 device and a Molly.jl fork
# that carries ext/MollyHIPExt.jl + ext/mhip_abi.jl and the stanzas of julia/Project.toml.fragment:
#
#     MOLLYHIP_ROOT=/path/to/this/checkout julia --project=/path/to/Molly.jl julia/test/runtests.jl
#
# What is restated (the CPU `Array` system is the oracle of every case, as the reference uses it against CuArray):
#   test/gpu_consistency.jl:3-50     33 atoms on a diagonal, box 20, r_c 5: forces / energy, rtol 1e-8, atol 1e-10
#   test/gpu_consistency.jl:52-114   100-atom lattice, spacing 1.5, σ = 1, r_c 4, through find_neighbors
#   test/gpu_consistency.jl:339-405  10 atoms, excluded (1,2), (2,3), special (1,3)
#   coupling.jl:861-932, README.md:126-133  the boundary replaced on a live system: scale_coords!, a move taken back, Langevin + MonteCarloBarostat
#   test/gpu_consistency.jl:407-449  20 atoms without a neighbour list
#   test/protein.jl:263-276          6mrr per-term forces / energies against the OpenMM files of the reference's data/ directory
using Test, LinearAlgebra, DelimitedFiles
using Molly, StaticArrays, Unitful
using AMDGPU

const T64 = Float64
lj_atoms(n, σ) = [Atom(index=i, mass=T64(1), charge=T64(0), σ=T64(σ), ϵ=T64(1)) for i in 1:n]
diagonal(n) = [SVector{3, T64}(0.5i, 0.5i, 0.5i) for i in 1:n]

"forces and energy of `gpu` (a ROCArray system, through libmollyhip) against `cpu` (the same system on Arrays), at the reference's tolerances"
function same_forces_and_energy(gpu, cpu; with_list::Bool)
    nb_cpu = cpu.neighbor_finder isa NoNeighborFinder ? nothing : find_neighbors(cpu)
    f_cpu = isnothing(nb_cpu) ? forces(cpu) : forces(cpu, nb_cpu)
    e_cpu = isnothing(nb_cpu) ? potential_energy(cpu) : potential_energy(cpu, nb_cpu)
    nb_gpu = with_list ? find_neighbors(gpu) : nothing
    f_gpu = Array(forces(gpu, nb_gpu))
    e_gpu = potential_energy(gpu, nb_gpu)
    @test all(isapprox(f_gpu[i], f_cpu[i]; rtol=1e-8, atol=1e-10) for i in eachindex(f_cpu))
    @test isapprox(e_gpu, e_cpu; rtol=1e-8, atol=1e-10)
end

function lj_pair(coords, atoms, boundary, r_cut; excluded=Tuple{Int, Int}[], special=Tuple{Int, Int}[], use_list=true)
    n = length(atoms)
    inter = (LennardJones(use_neighbors=use_list, cutoff=DistanceCutoff(T64(r_cut))),)
    common = (boundary=boundary, pairwise_inters=inter, force_units=NoUnits, energy_units=NoUnits)
    if use_list
        nf_gpu = GPUNeighborFinder(n_atoms=n, dist_cutoff=T64(r_cut), excluded_pairs=excluded, special_pairs=special, device_vector_type=ROCArray{Int32, 1})
        eligible = trues(n, n); spec = falses(n, n)
        for (i, j) in excluded; eligible[i, j] = eligible[j, i] = false; end
        for (i, j) in special; spec[i, j] = spec[j, i] = true; eligible[i, j] = eligible[j, i] = true; end      # (a special pair is listed, flagged)
        nf_cpu = DistanceNeighborFinder(eligible=eligible, special=spec, dist_cutoff=T64(r_cut))
    else
        nf_gpu = NoNeighborFinder(); nf_cpu = NoNeighborFinder()
    end
    gpu = System(; atoms=ROCArray(atoms), coords=ROCArray(coords), neighbor_finder=nf_gpu, common...)
    cpu = System(; atoms=atoms, coords=coords, neighbor_finder=nf_cpu, common...)
    return gpu, cpu
end

@testset "MollyHIPExt against Molly's CPU path" begin
    if !AMDGPU.functional()
        @warn "no functional AMDGPU device: nothing to test"
    else
        @test Molly.uses_gpu_neighbor_finder(ROCArray)                                  # the extension is loaded (≙ ext/MollyCUDAExt.jl:73)

        @testset "33 atoms on a diagonal" begin                                        # no cancellation between pair forces
            gpu, cpu = lj_pair(diagonal(33), lj_atoms(33, 0.3), CubicBoundary(T64(20)), 5.0)
            same_forces_and_energy(gpu, cpu; with_list=false)
        end
        @testset "100-atom lattice through find_neighbors" begin
            side = ceil(Int, 100^(1 / 3)); a = T64(1.5)
            coords = [SVector{3, T64}(i * a, j * a, k * a) for i in 1:side for j in 1:side for k in 1:side][1:100]
            gpu, cpu = lj_pair(coords, lj_atoms(100, 1.0), CubicBoundary(T64((side + 2) * a)), 4.0)
            same_forces_and_energy(gpu, cpu; with_list=true)
        end
        @testset "sys.boundary replaced on a live system" begin                          # scale_coords! (spatial.jl:1184-1210) as the barostats call it (coupling.jl:861-932)
            side = ceil(Int, 100^(1 / 3)); a = T64(1.5)
            coords = [SVector{3, T64}(i * a, j * a, k * a) for i in 1:side for j in 1:side for k in 1:side][1:100]
            gpu, cpu = lj_pair(coords, lj_atoms(100, 1.0), CubicBoundary(T64((side + 2) * a)), 4.0)
            same_forces_and_energy(gpu, cpu; with_list=true)                             # the engine's context and lists exist, on the old box
            old_boundary, old_coords = gpu.boundary, copy(gpu.coords)
            μ = SMatrix{3, 3, T64}(1.02, 0, 0, 0, 1.02, 0, 0, 0, 1.02)
            scale_coords!(gpu, μ); scale_coords!(cpu, μ)
            @test gpu.boundary != old_boundary
            same_forces_and_energy(gpu, cpu; with_list=true)                             # mhip_set_box behind context!(sys): the new box
            gpu.coords .= old_coords; gpu.boundary = old_boundary                        # a rejected Monte-Carlo move (coupling.jl:929-930)
            cpu.coords .= Array(old_coords); cpu.boundary = old_boundary
            same_forces_and_energy(gpu, cpu; with_list=true)
            # the README's GPU example in small (README.md:126-133): the stock Langevin loop with a MonteCarloBarostat, every force and trial energy from the engine
            temp = T64(1.0)
            sim = Langevin(dt=T64(0.002), temperature=temp, friction=T64(1.0), coupling=MonteCarloBarostat(T64(1.0), temp, gpu.boundary; n_steps=5))
            random_velocities!(gpu, temp)
            simulate!(gpu, sim, 20)
            @test all(isfinite, reduce(vcat, Array(gpu.coords)))
            @test sim.coupling.n_attempted == 4
            # … and through the device-resident stepper: the run is cut at the barostat's steps, apply_coupling! runs between two chunks, the engine follows the box
            ext = Base.get_extension(Molly, :MollyHIPExt)
            baro = MonteCarloBarostat(T64(1.0), temp, gpu.boundary; n_steps=5)
            simulate!(gpu, ext.HIPLangevin(dt=T64(0.002), temperature=temp, friction=T64(1.0), coupling=baro), 20)
            @test baro.n_attempted == 4 && all(isfinite, reduce(vcat, Array(gpu.coords)))
        end
        @testset "exclusions and a special pair" begin
            gpu, cpu = lj_pair(diagonal(10), lj_atoms(10, 0.3), CubicBoundary(T64(10)), 5.0; excluded=[(1, 2), (2, 3)], special=[(1, 3)])
            same_forces_and_energy(gpu, cpu; with_list=false)
        end
        @testset "no neighbour list" begin                                               # use_neighbors = false → NoNeighborList (force.jl:1219-1224): the engine's all-pairs context
            gpu, cpu = lj_pair(diagonal(20), lj_atoms(20, 0.3), CubicBoundary(T64(10)), 5.0; use_list=false)
            same_forces_and_energy(gpu, cpu; with_list=false)
            ext = Base.get_extension(Molly, :MollyHIPExt)
            @test haskey(ext.CONTEXTS_NOLIST, gpu) && !haskey(ext.CONTEXTS, gpu)         # it was the engine that answered, not the generic KernelAbstractions method
        end
        @testset "remove_CM_motion! on the device" begin                                # ≙ ext/MollyCUDAExt.jl:2373 against spatial.jl:920
            gpu, cpu = lj_pair(diagonal(33), lj_atoms(33, 0.3), CubicBoundary(T64(20)), 5.0)
            v = [SVector{3, T64}(0.1i, -0.2, 0.05i) for i in 1:33]
            gpu.velocities = ROCArray(v); cpu.velocities = copy(v)
            remove_CM_motion!(gpu); remove_CM_motion!(cpu)
            @test maximum(norm.(Array(gpu.velocities) .- cpu.velocities)) < 1e-12
        end

        @testset "6mrr against the OpenMM files" begin                                 # bars of test/protein.jl:263-276
            data = normpath(joinpath(dirname(pathof(Molly)), "..", "data"))
            ff = MolecularForceField(T64, joinpath.(data, "force_fields", ["ff99SBildn.xml", "tip3p_standard.xml"])...)
            mk(AT, method) = System(joinpath(data, "6mrr_equil.pdb"), ff; array_type=AT, center_coords=false, nonbonded_method=method, approximate_pme=false)
            for (method, tag) in ((:cutoff, "all_cut"), (:pme, "all_pme_exact"))
                sys = mk(ROCArray, method)
                f = Array(forces(sys))
                f_ref = SVector{3}.(eachrow(readdlm(joinpath(data, "openmm_6mrr", "amber", "forces_$tag.txt"))))u"kJ * mol^-1 * nm^-1"
                @test maximum(norm.(f .- f_ref)) < 1e-7u"kJ * mol^-1 * nm^-1"
                e_ref = readdlm(joinpath(data, "openmm_6mrr", "amber", "energy_$tag.txt"))[1] * u"kJ * mol^-1"
                @test abs(potential_energy(sys) - e_ref) < 1e-5u"kJ * mol^-1"
            end
            # 100 steps of 0.5 fs from OpenMM's start velocities through the device-resident stepper (test/protein.jl:278-299: 1e-10 nm, 1e-7 nm/ps)
            sys = mk(ROCArray, :pme)
            rd(name) = SVector{3}.(eachrow(readdlm(joinpath(data, "openmm_6mrr", name))))
            sys.velocities = ROCArray(rd("velocities_300K.txt")u"nm * ps^-1")
            ext = Base.get_extension(Molly, :MollyHIPExt)
            simulate!(sys, ext.HIPVelocityVerlet(dt=0.0005u"ps"), 100)
            x_ref = wrap_coords.(rd(joinpath("amber", "coordinates_100steps.txt"))u"nm", (sys.boundary,))
            @test maximum(norm.(Array(sys.coords) .- x_ref)) < 1e-10u"nm"
            @test maximum(norm.(Array(sys.velocities) .- rd(joinpath("amber", "velocities_100steps.txt"))u"nm * ps^-1")) < 1e-7u"nm * ps^-1"
        end
    end
end

@testset "MollyHIP (plugin route, unmodified Molly)" begin
    pkg = joinpath(@__DIR__, "..", "MollyHIP")
    if !AMDGPU.functional() || !isdir(pkg)
        @warn "skipped"
    else
        push!(LOAD_PATH, pkg); @eval using MollyHIP
        coords, atoms, b = diagonal(33), lj_atoms(33, 0.3), CubicBoundary(T64(20))
        lj = (LennardJones(use_neighbors=true, cutoff=DistanceCutoff(T64(5))),)
        cpu = System(atoms=atoms, coords=coords, boundary=b, pairwise_inters=lj, neighbor_finder=DistanceNeighborFinder(eligible=trues(33, 33), dist_cutoff=T64(5)),
                     force_units=NoUnits, energy_units=NoUnits)
        eng = MollyHIP.HIPNonbonded(lj; dist_cutoff=T64(5))
        # (a) the engine computes the forces as a general interaction of an Array system (host pointers)
        viaeng = System(atoms=atoms, coords=coords, boundary=b, general_inters=(eng,), force_units=NoUnits, energy_units=NoUnits)
        nb = find_neighbors(cpu)
        @test all(isapprox.(forces(viaeng), forces(cpu, nb); rtol=1e-8, atol=1e-10))
        @test isapprox(potential_energy(viaeng), potential_energy(cpu, nb); rtol=1e-8, atol=1e-10)
        # (b) the engine only finds the neighbours: the SAME pair set as Molly's own finder (neighbors.jl:409-411), Molly's CPU loop computes
        vianf = System(atoms=atoms, coords=coords, boundary=b, pairwise_inters=lj, neighbor_finder=MollyHIP.HIPNeighborFinder(eng), force_units=NoUnits, energy_units=NoUnits)
        nb_eng = find_neighbors(vianf)
        key(t) = (min(t[1], t[2]), max(t[1], t[2]), t[3])
        @test sort([key(nb_eng.list[k]) for k in 1:nb_eng.n]) == sort([key(nb.list[k]) for k in 1:nb.n])
        MollyHIP.release!(viaeng); MollyHIP.release!(vianf)
    end
end

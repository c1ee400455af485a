// step_fused.h — two launches of the complete PME step that each run TWO independent jobs side by side.
//
// A 16k-atom system (6mrr) is latency-bound: every kernel of the step lasts 5–35 µs against a ≈ 4.5 µs floor, the jobs are too small to
// fill 256 CUs, and they queue behind each other on the stream (separate streams cost more in cross-queue waits than they win, and
// any-order launches are not honoured on gfx9: DESIGN §4).  Jobs that do not depend on each other and fit the same workgroup shape
// therefore share a launch — the first workgroups do one job, the rest the other, and the hardware runs them at the same time:
//   k_spread_bonded   : charge spreading (ewald.jl:598-621)  ‖  bonds, angles, torsions, Ewald exclusions into their slots (kernels.jl:233-342)
//   k_gather_collect  : force interpolation (ewald.jl:805-840) ‖  per-atom sums of those slots
// The interpolation adds to the force array; the slot sums go to a side array that the integrator (or fold_side_forces) adds, so the
// two halves of the second launch never write the same word.
#pragma once
#include "bonded.h"
#include "pme.h"

namespace mhip {

template <class T, int ORDER, int PME_SB>
__global__ void __launch_bounds__(256) k_spread_bonded(int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, T* rgrid, PmeP<T> P, int n_spread, BondedArgs<T> B) {
    if ((int)blockIdx.x < n_spread) { pme_spread_blocks<T, ORDER, PME_SB>((int)blockIdx.x, n_spread, n_atoms, pos, rgrid, P); return; }
    double e = 0;
    bonded_terms<T, false, true>(B, ((int)blockIdx.x - n_spread) * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), 64, e);   // four 64-lane term blocks per workgroup
}

template <class T, int ORDER, bool STORE = false>
__global__ void __launch_bounds__(256) k_gather_collect(int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, const T* __restrict__ phi, typename Vec<T>::T4* frc, PmeP<T> P,
                                                        int n_gather, const int32_t* __restrict__ orig, const int32_t* __restrict__ role_start, const int32_t* __restrict__ role_slot,
                                                        const typename Vec<T>::T4* __restrict__ slots, typename Vec<T>::T4* side,
                                                        const typename Vec<T>::T4* __restrict__ parts, int n_parts, int64_t part_stride) {
    if ((int)blockIdx.x < n_gather) { pme_gather_blocks<T, ORDER, STORE>((int)blockIdx.x, n_gather, n_atoms, pos, phi, frc, P); return; }
    bonded_collect_lane<T, true>(((int64_t)blockIdx.x - n_gather) * blockDim.x + threadIdx.x, n_atoms, orig, role_start, role_slot, slots, side, parts, n_parts, part_stride);
}

// reciprocal-space PME forces added to frc (store: written to frc, every owned atom), bonded forces left in `side` (every owned atom written)
template <class T>
inline void launch_pme_bonded_fused(hipStream_t s, Pme<T>& pme, Bonded<T>& bonded, const GridP<T>& G, const InterP<T>& I, int64_t n_owned, int64_t cap,
                                    const typename Vec<T>::T4* pos, const int32_t* inv, const int32_t* orig, typename Vec<T>::T4* frc, typename Vec<T>::T4* side, bool store = false,
                                    bool spread_done = false) {      // spread_done: the charges are on the mesh and the terms in their slots already (forces_gs.hip's fused launch)
    bonded.ensure_roles(s, cap);
    const BondedArgs<T> B = bonded.slot_args(G, I, pos, inv);
    const int n_term_wg = cdiv(bonded.n_blocks(), 4);
    static const int sb = [] { const char* v = std::getenv("MOLLYHIP_PME_SPREAD_BATCH"); return v && *v ? std::atoi(v) : 64; }();   // atoms per spreading batch
    const int n_spread = (int)std::min<int64_t>(cdiv(n_owned, (int64_t)(sb <= 16 ? 16 : sb <= 32 ? 32 : 64)), 4096);
    auto spread = [&](auto order_tag) {
        constexpr int ORDER = decltype(order_tag)::value;
        if (sb <= 16) hipLaunchKernelGGL((k_spread_bonded<T, ORDER, 16>), dim3(n_spread + n_term_wg), dim3(256), 0, s, n_owned, pos, pme.rgrid.p, pme.P, n_spread, B);
        else if (sb <= 32) hipLaunchKernelGGL((k_spread_bonded<T, ORDER, 32>), dim3(n_spread + n_term_wg), dim3(256), 0, s, n_owned, pos, pme.rgrid.p, pme.P, n_spread, B);
        else hipLaunchKernelGGL((k_spread_bonded<T, ORDER, 64>), dim3(n_spread + n_term_wg), dim3(256), 0, s, n_owned, pos, pme.rgrid.p, pme.P, n_spread, B);
    };
    if (!spread_done) { if (pme.order == 4) spread(std::integral_constant<int, 4>{}); else if (pme.order == 5) spread(std::integral_constant<int, 5>{}); else spread(std::integral_constant<int, 6>{}); }
    pme.mesh_to_potential(s, nullptr, true);
    const int n_gather = (int)Pme<T>::atom_blocks(n_owned);
    const int n_collect = (int)cdiv(n_owned * COLLECT_LANES, (int64_t)256);
    auto gather = [&](auto order_tag) {
        constexpr int ORDER = decltype(order_tag)::value;
        if (store) hipLaunchKernelGGL((k_gather_collect<T, ORDER, true>), dim3(n_gather + n_collect), dim3(256), 0, s, n_owned, pos, (const T*)pme.phi.p, frc, pme.P, n_gather, orig,
                           (const int32_t*)bonded.role_start.p, (const int32_t*)bonded.role_slot.p, (const typename Vec<T>::T4*)bonded.slots, side, bonded.fold_parts, bonded.fold_n, bonded.fold_stride);
        else hipLaunchKernelGGL((k_gather_collect<T, ORDER>), dim3(n_gather + n_collect), dim3(256), 0, s, n_owned, pos, (const T*)pme.phi.p, frc, pme.P, n_gather, orig,
                           (const int32_t*)bonded.role_start.p, (const int32_t*)bonded.role_slot.p, (const typename Vec<T>::T4*)bonded.slots, side, bonded.fold_parts, bonded.fold_n, bonded.fold_stride);
    };
    if (pme.order == 4) gather(std::integral_constant<int, 4>{}); else if (pme.order == 5) gather(std::integral_constant<int, 5>{}); else gather(std::integral_constant<int, 6>{});
    bonded.fold_parts = nullptr; bonded.fold_n = 0;      // (consumed)
    MHIP_HIP(hipGetLastError());
}

}  // namespace mhip

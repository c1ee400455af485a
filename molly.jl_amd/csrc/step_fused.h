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
#include "kernels.h"
#include "philox.h"
#include "pme.h"

namespace mhip {

template <class T, int ORDER, int PME_SB>
__global__ void __launch_bounds__(256) k_spread_bonded(int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, T* rgrid, PmeP<T> P, int n_spread, BondedArgs<T> B) {
    if ((int)blockIdx.x < n_spread) { pme_spread_blocks<T, ORDER, PME_SB>((int)blockIdx.x, n_spread, n_atoms, pos, rgrid, P); return; }
    double e = 0;
    bonded_terms<T, false, true>(B, ((int)blockIdx.x - n_spread) * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), 64, e);   // four 64-lane term blocks per workgroup
}

template <class T, int ORDER>
__global__ void __launch_bounds__(256) k_gather_collect(int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, const T* __restrict__ phi, typename Vec<T>::T4* frc, PmeP<T> P,
                                                        int n_gather, const int32_t* __restrict__ orig, const int32_t* __restrict__ role_start, const int32_t* __restrict__ role_slot,
                                                        const typename Vec<T>::T4* __restrict__ slots, typename Vec<T>::T4* side,
                                                        const typename Vec<T>::T4* __restrict__ parts, int n_parts, int64_t part_stride) {
    if ((int)blockIdx.x < n_gather) { pme_gather_blocks<T, ORDER>((int)blockIdx.x, n_gather, n_atoms, pos, phi, frc, P); return; }
    bonded_collect_lane<T, true>(((int64_t)blockIdx.x - n_gather) * blockDim.x + threadIdx.x, n_atoms, orig, role_start, role_slot, slots, side, parts, n_parts, part_stride);
}

// The last force launch of a small system's step WITH the velocity-Verlet update in it (round 5; mid-run steps of mhip_vv_run).  k_gather_collect leaves the
// reciprocal-space force in frc and the bonded sums in a side array, and k_vv_mid — 5.6 µs for 16 k atoms, nearly all of it launch and latency — adds the two and
// integrates.  Here one workgroup does all of it for its batch of PME_AB atoms: the bonded slot sums (eight lanes per atom, asked for first: their index → record
// chain runs while the spline tables are made), the interpolation (a half-wave per atom), then ONE lane per atom adds pair force + reciprocal force + bonded
// force in k_vv_mid's order and runs k_vv_mid's arithmetic (same helpers: second kick, Σ m v partial, first kick, drift, wrap; v_cm of the step before, one launch
// late, from the ONE partial the pair launch's extra workgroup left).  Coordinates are updated in place: nothing in this launch reads another atom's.
template <class T> struct GcvArgs {
    int64_t n_atoms; typename Vec<T>::T4* pos; typename Vec<T>::T4* vel; const typename Vec<T>::T4* frc; const T* phi; PmeP<T> P;
    const int32_t* orig; const int32_t* role_start; const int32_t* role_slot; const typename Vec<T>::T4* slots; const typename Vec<T>::T4* parts; int n_parts; int64_t part_stride;
    T dt, dt2; const double* cm_in; double* cm_out; GridP<T> G;
    const typename Vec<T>::T4* snap_a; const typename Vec<T>::T4* snap_b; float* trk_part;      // validity check of the pair lists, as in k_vv_mid (nullable)
    // LANG instantiations (mhip_langevin_run, round 6): the integrating lane runs the Langevin-middle update of stochastic.hip instead (philox.h, langevin_atom: the same
    // function k_langevin calls); v_cm of the step before is subtracted from the velocity only, as k_langevin does (the reference removes it behind the step's drift)
    StochP<T> S;
};
template <class T, int ORDER, bool LANG = false>
__global__ void __launch_bounds__(256) k_gather_collect_vv(GcvArgs<T> A) {
    using T4 = typename Vec<T>::T4;
    __shared__ T l_w[6 * ORDER * PME_AB]; __shared__ int l_i[3 * PME_AB]; __shared__ T l_q[PME_AB]; __shared__ T l_g[3 * PME_AB];
    __shared__ double l_cm[PME_AB][4]; __shared__ float l_tr[PME_AB][3];
    static_assert(COLLECT_LANES * PME_AB <= 256, "the batch's slot sums take the first COLLECT_LANES * PME_AB lanes of the workgroup");
    const int tid = threadIdx.x, sub = tid & 31, hw = tid >> 5, ta = tid / COLLECT_LANES;
    const bool grp = tid < COLLECT_LANES * PME_AB;                       // (whole waves)
    T vc[3] = {T(0), T(0), T(0)};
    if (A.cm_in) { const double m = A.cm_in[3]; for (int c = 0; c < 3; ++c) vc[c] = (T)(A.cm_in[c] / m); }      // (block_vcm over one partial)
    const T sh[3] = {M<T>::mul(vc[0], A.dt), M<T>::mul(vc[1], A.dt), M<T>::mul(vc[2], A.dt)};
    double px = 0, py = 0, pz = 0, pm = 0;
    float v2m = 0.f, dam = 0.f, dbm = 0.f;
    for (int64_t a0 = (int64_t)blockIdx.x * PME_AB; a0 < A.n_atoms; a0 += (int64_t)gridDim.x * PME_AB) {
        T bx = T(0), by = T(0), bz = T(0);
        bool mine = false, any = false;
        T4 v = make4<T>(T(0), T(0), T(0), T(1)), f = v, p = v, qa = v, qb = v;
        const int64_t s = a0 + ta;
        if (a0 != (int64_t)blockIdx.x * PME_AB) __syncthreads();      // (a further batch: the tables of the one before are done with)
        if (grp) {
            mine = bonded_collect_sum<T>(a0 * COLLECT_LANES + tid, A.n_atoms, A.orig, A.role_start, A.role_slot, A.slots, A.parts, A.n_parts, A.part_stride, bx, by, bz, any);
            if (mine) { v = A.vel[s]; f = A.frc[s]; p = A.pos[s]; if (A.trk_part) { qa = A.snap_a[s]; qb = A.snap_b[s]; } }
        }
        // (the spline tables by the LAST wave, while the first two wait for their slot records: two independent chains of memory latencies side by side)
        pme_atom_tables<T, ORDER, true>(a0, A.n_atoms, A.pos, A.P, l_w, l_i, l_q, 192);
        __syncthreads();
        for (int t = hw; t < PME_AB; t += 8) {
            T sx, sy, sz;
            pme_gather_one<T, ORDER>(t, sub, l_q[t], l_w, l_i, A.phi, A.P, sx, sy, sz);
            if (sub == 0) { l_g[3 * t] = sx; l_g[3 * t + 1] = sy; l_g[3 * t + 2] = sz; }
        }
        __syncthreads();
        if (mine) {
            const T q = l_q[ta];
            if (q != T(0)) { f.x -= q * l_g[3 * ta]; f.y -= q * l_g[3 * ta + 1]; f.z -= q * l_g[3 * ta + 2]; }      // (k_gather_collect: frc += reciprocal part)
            f.x += bx; f.y += by; f.z += bz;                                                                          // (k_vv_mid: + the bonded sums)
            if constexpr (LANG) {
                if (A.cm_in) { v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2]; }
                langevin_atom<T>(v, p, f, A.S, (uint64_t)A.orig[s] + 1, A.G);
                if (A.cm_out) { px += (double)v.x * v.w; py += (double)v.y * v.w; pz += (double)v.z * v.w; pm += v.w; }
            } else {
                if (A.cm_in) {
                    v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2];
                    p.x = M<T>::sub(p.x, sh[0]); p.y = M<T>::sub(p.y, sh[1]); p.z = M<T>::sub(p.z, sh[2]);
                }
                const T kx = M<T>::mul(accel_of(f.x, v.w), A.dt2), ky = M<T>::mul(accel_of(f.y, v.w), A.dt2), kz = M<T>::mul(accel_of(f.z, v.w), A.dt2);
                v.x = M<T>::add(v.x, kx); v.y = M<T>::add(v.y, ky); v.z = M<T>::add(v.z, kz);
                if (A.cm_out) { px += (double)v.x * v.w; py += (double)v.y * v.w; pz += (double)v.z * v.w; pm += v.w; }
                v.x = M<T>::add(v.x, kx); v.y = M<T>::add(v.y, ky); v.z = M<T>::add(v.z, kz);
                p.x = step_add(p.x, v.x, A.dt); p.y = step_add(p.y, v.y, A.dt); p.z = step_add(p.z, v.z, A.dt);
                wrap_point(p.x, p.y, p.z, A.G);
            }
            A.pos[s] = p; A.vel[s] = v;
            if (A.trk_part) {
                v2m = fmaxf(v2m, (float)(v.x * v.x + v.y * v.y + v.z * v.z));
                T ex = p.x - qa.x, ey = p.y - qa.y, ez = p.z - qa.z;
                disp_image(ex, ey, ez, A.G);
                dam = fmaxf(dam, (float)(ex * ex + ey * ey + ez * ez));
                ex = p.x - qb.x; ey = p.y - qb.y; ez = p.z - qb.z;
                disp_image(ex, ey, ez, A.G);
                dbm = fmaxf(dbm, (float)(ex * ex + ey * ey + ez * ez));
            }
        }
    }
    // the workgroup's Σ m v partial and the three maxima: its PME_AB integrating lanes (lane 0 of every group of COLLECT_LANES) in atom order
    if (grp && (tid % COLLECT_LANES) == 0) { l_cm[ta][0] = px; l_cm[ta][1] = py; l_cm[ta][2] = pz; l_cm[ta][3] = pm; l_tr[ta][0] = dam; l_tr[ta][1] = dbm; l_tr[ta][2] = v2m; }
    __syncthreads();
    if (A.cm_out && tid < 4) { double a = 0; for (int t = 0; t < PME_AB; ++t) a += l_cm[t][tid]; A.cm_out[4 * (int64_t)blockIdx.x + tid] = a; }
    if (A.trk_part && tid < 3) { float m = 0.f; for (int t = 0; t < PME_AB; ++t) m = fmaxf(m, l_tr[t][tid]); A.trk_part[tid * gridDim.x + blockIdx.x] = m; }
}

// reciprocal-space PME forces added to frc, bonded forces left in `side` (every owned atom written)
template <class T>
inline void launch_pme_bonded_fused(hipStream_t s, Pme<T>& pme, Bonded<T>& bonded, const GridP<T>& G, const InterP<T>& I, int64_t n_owned, int64_t cap,
                                    const typename Vec<T>::T4* pos, const int32_t* inv, const int32_t* orig, typename Vec<T>::T4* frc, typename Vec<T>::T4* side,
                                    bool spread_done = false,        // spread_done: the charges are on the mesh and the terms in their slots already (forces_gs.hip's fused launch)
                                    const GcvArgs<T>* vv = nullptr,        // vv: the last launch integrates (its dt / v_cm / partial / tracking fields filled in by the caller); frc then holds the pair forces and is only read
                                    bool lang = false) {                   // … with the Langevin-middle update (vv->S) instead of the velocity-Verlet one
    bonded.ensure_roles(s, cap);
    const BondedArgs<T> B = bonded.slot_args(G, I, pos, inv);
    const int n_term_wg = cdiv(bonded.n_blocks(), 4);
    const int n_spread = (int)std::min<int64_t>(cdiv(n_owned, (int64_t)64), 4096);      // 64 atoms per spreading batch
    auto spread = [&](auto order_tag) {
        constexpr int ORDER = decltype(order_tag)::value;
        hipLaunchKernelGGL((k_spread_bonded<T, ORDER, 64>), dim3(n_spread + n_term_wg), dim3(256), 0, s, n_owned, pos, pme.rgrid.p, pme.P, n_spread, B);
    };
    if (!spread_done) { if (pme.order == 4) spread(std::integral_constant<int, 4>{}); else if (pme.order == 5) spread(std::integral_constant<int, 5>{}); else spread(std::integral_constant<int, 6>{}); }
    pme.mesh_to_potential(s, nullptr, true);
    const int n_gather = (int)Pme<T>::atom_blocks(n_owned);
    const int n_collect = (int)cdiv(n_owned * COLLECT_LANES, (int64_t)256);
    auto gather = [&](auto order_tag) {
        constexpr int ORDER = decltype(order_tag)::value;
        hipLaunchKernelGGL((k_gather_collect<T, ORDER>), dim3(n_gather + n_collect), dim3(256), 0, s, n_owned, pos, (const T*)pme.phi.p, frc, pme.P, n_gather, orig,
                           (const int32_t*)bonded.role_start.p, (const int32_t*)bonded.role_slot.p, (const typename Vec<T>::T4*)bonded.slots, side, bonded.fold_parts, bonded.fold_n, bonded.fold_stride);
    };
    if (vv) {
        GcvArgs<T> V = *vv;
        V.n_atoms = n_owned; V.pos = const_cast<typename Vec<T>::T4*>(pos); V.frc = frc; V.phi = (const T*)pme.phi.p; V.P = pme.P;
        V.orig = orig; V.role_start = (const int32_t*)bonded.role_start.p; V.role_slot = (const int32_t*)bonded.role_slot.p; V.slots = (const typename Vec<T>::T4*)bonded.slots;
        V.parts = bonded.fold_parts; V.n_parts = bonded.fold_n; V.part_stride = bonded.fold_stride;
        if (lang) {
            if (pme.order == 4) hipLaunchKernelGGL((k_gather_collect_vv<T, 4, true>), dim3(n_gather), dim3(256), 0, s, V);
            else if (pme.order == 5) hipLaunchKernelGGL((k_gather_collect_vv<T, 5, true>), dim3(n_gather), dim3(256), 0, s, V);
            else hipLaunchKernelGGL((k_gather_collect_vv<T, 6, true>), dim3(n_gather), dim3(256), 0, s, V);
        }
        else if (pme.order == 4) hipLaunchKernelGGL((k_gather_collect_vv<T, 4>), dim3(n_gather), dim3(256), 0, s, V);
        else if (pme.order == 5) hipLaunchKernelGGL((k_gather_collect_vv<T, 5>), dim3(n_gather), dim3(256), 0, s, V);
        else hipLaunchKernelGGL((k_gather_collect_vv<T, 6>), dim3(n_gather), dim3(256), 0, s, V);
    }
    else if (pme.order == 4) gather(std::integral_constant<int, 4>{}); else if (pme.order == 5) gather(std::integral_constant<int, 5>{}); else gather(std::integral_constant<int, 6>{});
    bonded.fold_parts = nullptr; bonded.fold_n = 0;      // (consumed)
    MHIP_HIP(hipGetLastError());
}

}  // namespace mhip

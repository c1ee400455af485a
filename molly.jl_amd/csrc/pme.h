// pme.h — smooth particle-mesh Ewald, reciprocal-space term, single domain, CubicBoundary.
// Behavioural spec: src/interactions/ewald.jl — pme_bspline_moduli :311-358, grid_placement_inner! :484-493,
// update_bsplines_inner! :518-556, spread_charge_inner! :598-621, recip_conv_inner! :676-725,
// interpolate_force_inner! :805-840, ewald_pe_forces! :873-929.
//
// gfx950 design.  The reference's meshes are whatever pme_params (:479-482) yields — 46×46×51 for 6mrr, factors 23 and 17 —
// and are tiny (≈10⁵ points, < 1 MB: L2-resident), so the transform is latency-, not bandwidth- or flop-bound.  Instead of a
// library FFT (Bluestein chains of many small launches for such sizes) the three axis transforms are direct DFTs on
// LDS-staged line tiles: any n, one launch per axis, and the x pass does forward-x · influence function · backward-x in
// one kernel, so a whole reciprocal evaluation is 7 launches:
//   spread (LDS sub-mesh per atom batch, one float atomic per touched mesh point) → z real-to-complex → y → x·conv·x⁻¹ → y⁻¹
//   → z complex-to-real → gather (32 lanes per atom).
// Grid layout: complex<T> at ((x·ny + y)·nz + z), z fastest (the reference's charge_grid[z, y, x]).
#pragma once
#include <vector>

#include "physics.h"
#include "pme_fft.h"

namespace mhip {

template <class T> struct PmeP {
    int n[3];
    T invL[3], n_over_L[3];
    T f_div_er, factor, pi_V;        // ke/ϵr, π²/α², π·V
    int tri; T r[3][3];              // TriclinicBoundary: recip_box = invert_box_vectors(boundary) (spatial.jl:338-347), r[e][d] = recip_box[e+1][d+1]; lower triangular
};

// update_bsplines_inner! (:518-556) for one fractional offset: θ and dθ/du of the ORDER B-spline weights
template <class T, int ORDER> __device__ inline void pme_bspline(T dr, T* b, T* db) {
#pragma unroll
    for (int k = 0; k < ORDER; ++k) b[k] = T(0);
    b[1] = dr; b[0] = T(1) - dr;
#pragma unroll
    for (int k = 3; k <= ORDER - 1; ++k) {
        const T dv = T(1) / T(k - 1);
        b[k - 1] = dv * dr * b[k - 2];
#pragma unroll
        for (int l = 1; l <= k - 2; ++l) b[k - l - 1] = dv * ((dr + T(l)) * b[k - l - 2] + (T(k - l) - dr) * b[k - l - 1]);
        b[0] *= dv * (T(1) - dr);
    }
    db[0] = -b[0];
#pragma unroll
    for (int k = 1; k <= ORDER - 1; ++k) db[k] = b[k - 1] - b[k];
    const T dv = T(1) / T(ORDER - 1);
    b[ORDER - 1] = dv * dr * b[ORDER - 2];
#pragma unroll
    for (int l = 1; l <= ORDER - 2; ++l) b[ORDER - l - 1] = dv * ((dr + T(l)) * b[ORDER - l - 2] + (T(ORDER - l) - dr) * b[ORDER - l - 1]);
    b[0] *= dv * (T(1) - dr);
}
// grid_placement_inner! (:484-493): first mesh index and fractional offset along one axis
template <class T> __device__ inline T pme_frac(const T* c, int d, const PmeP<T>& P) {      // sum(coords[i] .* recip_box[:, d]) (:486)
    return P.tri ? c[0] * P.r[0][d] + c[1] * P.r[1][d] + c[2] * P.r[2][d] : c[d] * P.invL[d];
}
template <class T> __device__ inline void pme_place(T t, int n, int& idx, T& dr) {
    t = (t - M<T>::floor(t)) * T(n);
    const int ti = (int)M<T>::floor(t);
    dr = t - T(ti);
    idx = ti % n;
}
template <class T, int ORDER> __device__ inline T pick(const T* a, int i) {   // a[i] of a register array, no scratch
    T v = a[0];
#pragma unroll
    for (int k = 1; k < ORDER; ++k) v = (k == i) ? a[k] : v;
    return v;
}

constexpr int PME_AB = 16;       // atoms per block and round of the gather kernel (small batches: many short blocks; 64 -> 16 halved its time)

// Phase 1 of spread and gather: ONE lane per atom evaluates grid_placement_inner! and update_bsplines_inner! (the recursion is
// serial and identical for every mesh point of the atom) and parks first index, charge and the 3·ORDER weights (and
// derivatives) in LDS, [value][atom] so that both the writes (lanes = atoms) and the later reads are conflict-free.
template <class T, int ORDER, bool DERIV>
__device__ inline void pme_atom_tables(int64_t a0, int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, const PmeP<T>& P, T* l_w, int* l_i, T* l_q, int first_lane = 0) {
    const int t = (int)threadIdx.x - first_lane;      // (first_lane: which lanes of the workgroup make the tables — the first PME_AB unless those are busy with something else)
    if (t >= 0 && t < PME_AB) {
        const int64_t a = a0 + t;
        T q = T(0); int i0[3] = {0, 0, 0};
        T th[ORDER], dth[ORDER], dr;
        if (a < n_atoms) {
            const auto p = pos[a];
            q = p.w;
            const T c[3] = {p.x, p.y, p.z};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                pme_place<T>(pme_frac<T>(c, d, P), P.n[d], i0[d], dr);
                pme_bspline<T, ORDER>(dr, th, dth);
#pragma unroll
                for (int k = 0; k < ORDER; ++k) { l_w[((d * ORDER + k) * (DERIV ? 2 : 1)) * PME_AB + t] = th[k]; if constexpr (DERIV) l_w[((d * ORDER + k) * 2 + 1) * PME_AB + t] = dth[k]; }
            }
        }
        l_q[t] = q; l_i[t] = i0[0]; l_i[PME_AB + t] = i0[1]; l_i[2 * PME_AB + t] = i0[2];
    }
}

// spread_charge_inner! (:598-621).  A block takes PME_SB consecutive (Hilbert-sorted, hence spatially compact) atoms.
// Phase 1: one lane per atom (tables above).  Phase 2: a 32-lane half-wave per atom, lane ↔ (iy, iz), loop over ix, adds the
// ORDER³ weights into an LDS sub-mesh that covers the batch's bounding box (ds_add_f32; relative mesh indices, so the box may
// straddle the periodic boundary).  Phase 3: the sub-mesh is flushed with ONE global float atomic per touched mesh point —
// several times fewer than one per (atom, point).  A batch whose box does not fit the LDS sub-mesh (a jump of the curve) falls
// back to direct global atomics.  (One mesh per XCD, selected by the hardware XCC id, was measured too: the spread did not get
// faster and the first transform pass paid for summing and re-zeroing eight copies.)
// The LDS sub-mesh accumulates in 64-bit FIXED POINT: on gfx950 ds_add_f32 runs at a tenth of the rate of ds_add_u32 / ds_add_u64
// (tools/micro/lds_atomic_rate.hip: 8 000 adds of this access pattern per block take 22.6 µs as floats, 6.7 µs — launch included — as
// 64-bit integers), and the LDS adds were half of the spreading kernel.  A contribution q·w (|w| ≤ 0.3 for orders 4–6) is scaled by
// 2²⁸ (fp32: resolution 3.7e-9; a batch with a charge beyond 24 e takes the direct path) or 2⁴⁴ (fp64: 5.7e-14), rounded once, sign-extended and added as an unsigned 64-bit word
// (two's complement: no overflow within 2³⁵ such terms); the sums themselves are then exact and independent of the order of the adds.
constexpr int PME_BOX_BYTES = 48 * 1024;   // LDS sub-mesh, 8 bytes per point
template <class T> struct PmeFix;
template <> struct PmeFix<float> {
    static constexpr float scale = 268435456.f, inv = 1.f / 268435456.f;            // 2^28
    __device__ static inline unsigned long long enc(float v) { return (unsigned long long)(long long)__float2int_rn(v * scale); }
};
template <> struct PmeFix<double> {
    static constexpr double scale = 17592186044416.0, inv = 1.0 / 17592186044416.0;  // 2^44
    __device__ static inline unsigned long long enc(double v) { return (unsigned long long)__double2ll_rn(v * scale); }
};
#ifndef MHIP_SPREAD_EXP
#define MHIP_SPREAD_EXP 0                  // timing experiments: 1 = stop behind the B-splines, 2 = behind the zeroed sub-mesh, 3 = behind the LDS adds (no flush)
#endif

// bytes of LDS a spreading workgroup needs in front of its sub-mesh (DYN: carved from the launch's dynamic LDS)
template <class T, int ORDER, int PME_SB> constexpr int pme_spread_head_bytes() { return ((3 * ORDER * PME_SB + PME_SB) * (int)sizeof(T) + (3 * PME_SB + 8) * (int)sizeof(int) + 15) & ~15; }
// DYN: the tables and the sub-mesh live in `dyn_lds` (dyn_bytes of it), not in static arrays — for a launch that shares its workgroups with
// jobs that have an LDS carve-up of their own (forces_gs.hip: the pair pass of a small system beside the spreading)
template <class T, int ORDER, int PME_SB, bool DYN = false>
__device__ inline void pme_spread_blocks(int bid, int nblk, int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, T* rgrid, const PmeP<T>& P,
                                         unsigned char* dyn_lds = nullptr, int dyn_bytes = 0) {
    T* l_w; int* l_i; T* l_q; unsigned long long* l_box; int* l_lo; int* l_hi; int PME_BOX;
    if constexpr (DYN) {
        l_box = reinterpret_cast<unsigned long long*>(dyn_lds + pme_spread_head_bytes<T, ORDER, PME_SB>());
        PME_BOX = (dyn_bytes - pme_spread_head_bytes<T, ORDER, PME_SB>()) / 8;
        l_w = reinterpret_cast<T*>(dyn_lds); l_q = l_w + 3 * ORDER * PME_SB; l_i = reinterpret_cast<int*>(l_q + PME_SB); l_lo = l_i + 3 * PME_SB; l_hi = l_lo + 4;
    } else {
        __shared__ T s_w[3 * ORDER * PME_SB]; __shared__ int s_i[3 * PME_SB]; __shared__ T s_q[PME_SB];
        __shared__ unsigned long long s_box[PME_BOX_BYTES / 8]; __shared__ int s_lo[3], s_hi[3];
        l_w = s_w; l_i = s_i; l_q = s_q; l_box = s_box; l_lo = s_lo; l_hi = s_hi; PME_BOX = PME_BOX_BYTES / 8;
    }
    const int tid = threadIdx.x, sub = tid & 31, hw = tid >> 5;
    T* mesh = rgrid;
    for (int64_t a0 = (int64_t)bid * PME_SB; a0 < n_atoms; a0 += (int64_t)nblk * PME_SB) {
        __syncthreads();
        // phase 1 (one thread per atom)
        if (tid < PME_SB) {
            const int64_t a = a0 + tid;
            T q = T(0); int i0[3] = {0, 0, 0};
            if (a < n_atoms) {
                const auto p = pos[a];
                q = p.w;
                const T c[3] = {p.x, p.y, p.z};
                T th[ORDER], dth[ORDER], dr;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    pme_place<T>(pme_frac<T>(c, d, P), P.n[d], i0[d], dr);
                    pme_bspline<T, ORDER>(dr, th, dth);
#pragma unroll
                    for (int k = 0; k < ORDER; ++k) l_w[(d * ORDER + k) * PME_SB + tid] = th[k];
                }
            }
            l_q[tid] = q; l_i[tid] = i0[0]; l_i[PME_SB + tid] = i0[1]; l_i[2 * PME_SB + tid] = i0[2];
        }
        if (tid < 3) { l_lo[tid] = 1 << 30; l_hi[tid] = -(1 << 30); }
        __syncthreads();
        if constexpr (MHIP_SPREAD_EXP == 1) continue;
        // (a charge beyond the fixed-point range of the sub-mesh — 2²⁸·q·w must fit 32 bits in fp32 — sends its batch down the direct path)
        const bool q_fix_ok = __syncthreads_and(tid >= PME_SB || !(M<T>::fabs(l_q[tid]) > T(24)));
        // bounding box of the first indices, relative to the batch's first atom and folded into [−n/2, n/2)
        const int ref[3] = {l_i[0], l_i[PME_SB], l_i[2 * PME_SB]};
        int rel[3];
        const int ta = tid < PME_SB ? tid : 0;
        const bool mine = tid < PME_SB && l_q[ta] != T(0);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int r = l_i[d * PME_SB + ta] - ref[d];
            const int n = P.n[d], h = n >> 1;
            r += r < -h ? n : 0; r -= r >= n - h ? n : 0;
            rel[d] = r;
            int lo = mine ? r : (1 << 30), hi = mine ? r : -(1 << 30);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
            if ((tid & 63) == 0) { atomicMin(&l_lo[d], lo); atomicMax(&l_hi[d], hi); }
        }
        __syncthreads();
        const int lo3[3] = {l_lo[0], l_lo[1], l_lo[2]};
        const bool empty = l_hi[0] < lo3[0];
        const int ex = l_hi[0] - lo3[0] + ORDER, ey = l_hi[1] - lo3[1] + ORDER, ez = l_hi[2] - lo3[2] + ORDER;
        const bool fits = !empty && q_fix_ok && ex <= P.n[0] && ey <= P.n[1] && ez <= P.n[2] && (int64_t)ex * ey * ez <= PME_BOX;
        if (fits) {   // the atoms' offsets inside the box replace their absolute first indices (everybody has read `ref` by now)
            for (int c = tid; c < ex * ey * ez; c += 256) l_box[c] = 0ull;
            if (tid < PME_SB) { l_i[tid] = rel[0] - lo3[0]; l_i[PME_SB + tid] = rel[1] - lo3[1]; l_i[2 * PME_SB + tid] = rel[2] - lo3[2]; }
        }
        __syncthreads();
        if constexpr (MHIP_SPREAD_EXP == 2) continue;
        if (!empty) {
            // the two half-waves of a wave take atoms half a batch apart: neighbours in the sorted order overlap in the sub-mesh and
            // their ds_add_f32 would hit the same addresses in the same instruction
            for (int t = (hw >> 1) + (hw & 1) * (PME_SB / 2), k = 0; k < PME_SB / 8; ++k, t += 4) {
                const T q = l_q[t];
                if (q == T(0)) continue;                                   // also the atoms past the end
                const int bx = l_i[t], by = l_i[PME_SB + t], bz = l_i[2 * PME_SB + t];
                for (int pr = sub; pr < ORDER * ORDER; pr += 32) {
                    const int iy = pr / ORDER, iz = pr - iy * ORDER;
                    const T qyz = (q * l_w[(ORDER + iy) * PME_SB + t]) * l_w[(2 * ORDER + iz) * PME_SB + t];
                    T wx[ORDER];                       // all x weights first: the adds below then issue back to back
#pragma unroll
                    for (int ix = 0; ix < ORDER; ++ix) wx[ix] = l_w[ix * PME_SB + t] * qyz;
                    if (fits) {
                        unsigned long long* col = l_box + (by + iy) * ez + (bz + iz);
#pragma unroll
                        for (int ix = 0; ix < ORDER; ++ix) atomicAdd(col + (bx + ix) * ey * ez, PmeFix<T>::enc(wx[ix]));
                    } else {
                        int yi = by + iy; yi -= yi >= P.n[1] ? P.n[1] : 0;
                        int zi = bz + iz; zi -= zi >= P.n[2] ? P.n[2] : 0;
                        T* col = mesh + (int64_t)yi * P.n[2] + zi;
#pragma unroll
                        for (int ix = 0; ix < ORDER; ++ix) {
                            int xi = bx + ix; xi -= xi >= P.n[0] ? P.n[0] : 0;
                            atomicAdd(col + (int64_t)xi * P.n[1] * P.n[2], wx[ix]);
                        }
                    }
                }
            }
        }
        if constexpr (MHIP_SPREAD_EXP == 3) continue;
        if (fits) {
            __syncthreads();
            // phase 3: box cell (cx, cy, cz) is mesh point (first index of the batch's first atom + l_lo + c) mod n
            int base[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { int v = (ref[d] + lo3[d]) % P.n[d]; base[d] = v < 0 ? v + P.n[d] : v; }
            // a thread owns a (cy, cz) column of the box and walks it along x: one division pair per column instead of two per cell
            // (the divisors are run-time values: ≈ 70 instructions per cell, a quarter of the kernel at 6 000 cells per batch)
            const int pl = ey * ez;
            for (int pc = tid; pc < pl; pc += 256) {
                const int cy = pc / ez, cz = pc - cy * ez;
                int yi = base[1] + cy; yi -= yi >= P.n[1] ? P.n[1] : 0;
                int zi = base[2] + cz; zi -= zi >= P.n[2] ? P.n[2] : 0;
                T* col = mesh + (int64_t)yi * P.n[2] + zi;
                int xi = base[0];
                for (int cx = 0; cx < ex; ++cx) {
                    const long long w = (long long)l_box[cx * pl + pc];
                    if (w != 0) atomicAdd(col + (int64_t)xi * P.n[1] * P.n[2], (T)w * PmeFix<T>::inv);
                    ++xi; xi -= xi >= P.n[0] ? P.n[0] : 0;
                }
            }
        }
    }
}

template <class T, int ORDER, int PME_SB>
__global__ void __launch_bounds__(256) k_pme_spread(int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, T* rgrid, PmeP<T> P) {
    pme_spread_blocks<T, ORDER, PME_SB>((int)blockIdx.x, (int)gridDim.x, n_atoms, pos, rgrid, P);
}

// interpolate_force_inner! (:805-840): Fs[i] -= q (∂θ/∂r ⊗ θ ⊗ θ) · φ, same two phases, shuffle reduction inside the half-wave
// one atom of a staged batch by one 32-lane half-wave (lane ↔ (iy, iz), loop over ix): the force per unit charge, the same value in every lane of the half
template <class T, int ORDER>
__device__ inline void pme_gather_one(int t, int sub, T q, const T* l_w, const int* l_i, const T* __restrict__ grid, const PmeP<T>& P, T& sx, T& sy, T& sz) {
    const int i0x = l_i[t], i0y = l_i[PME_AB + t], i0z = l_i[2 * PME_AB + t];
    T fx = T(0), fy = T(0), fz = T(0);
    if (q != T(0)) {
        for (int pr = sub; pr < ORDER * ORDER; pr += 32) {
            const int iy = pr / ORDER, iz = pr - iy * ORDER;
            int yi = i0y + iy; yi -= yi >= P.n[1] ? P.n[1] : 0;
            int zi = i0z + iz; zi -= zi >= P.n[2] ? P.n[2] : 0;
            const T ty = l_w[((ORDER + iy) * 2) * PME_AB + t], dty = l_w[((ORDER + iy) * 2 + 1) * PME_AB + t];
            const T tz = l_w[((2 * ORDER + iz) * 2) * PME_AB + t], dtz = l_w[((2 * ORDER + iz) * 2 + 1) * PME_AB + t];
            const T tyz = ty * tz, dty_tz = dty * tz, ty_dtz = ty * dtz;
            const auto* col = grid + (int64_t)yi * P.n[2] + zi;
            T g[ORDER];
#pragma unroll
            for (int ix = 0; ix < ORDER; ++ix) {
                int xi = i0x + ix; xi -= xi >= P.n[0] ? P.n[0] : 0;
                g[ix] = col[(int64_t)xi * P.n[1] * P.n[2]];
            }
#pragma unroll
            for (int ix = 0; ix < ORDER; ++ix) {
                const T tx = l_w[(ix * 2) * PME_AB + t], dtx = l_w[(ix * 2 + 1) * PME_AB + t];
                fx += dtx * tyz * g[ix]; fy += tx * dty_tz * g[ix]; fz += tx * ty_dtz * g[ix];
            }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { fx += __shfl_xor(fx, o, 64); fy += __shfl_xor(fy, o, 64); fz += __shfl_xor(fz, o, 64); }   // stays inside the 32-lane half
    // mesh derivatives → Cartesian force per unit charge: (fx·nx, fy·ny, fz·nz) · recip_box (:846-849)
    sx = fx * P.n_over_L[0]; sy = fy * P.n_over_L[1]; sz = fz * P.n_over_L[2];
    if (P.tri) {
        const T gx = fx * T(P.n[0]), gy = fy * T(P.n[1]), gz = fz * T(P.n[2]);
        sx = gx * P.r[0][0]; sy = gx * P.r[1][0] + gy * P.r[1][1]; sz = gx * P.r[2][0] + gy * P.r[2][1] + gz * P.r[2][2];
    }
}

template <class T, int ORDER>
__device__ inline void pme_gather_blocks(int bid, int nblk, int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, const T* __restrict__ grid,
                                         typename Vec<T>::T4* frc, const PmeP<T>& P) {
    __shared__ T l_w[6 * ORDER * PME_AB]; __shared__ int l_i[3 * PME_AB]; __shared__ T l_q[PME_AB];
    const int tid = threadIdx.x, sub = tid & 31, hw = tid >> 5;
    for (int64_t a0 = (int64_t)bid * PME_AB; a0 < n_atoms; a0 += (int64_t)nblk * PME_AB) {
        __syncthreads();
        pme_atom_tables<T, ORDER, true>(a0, n_atoms, pos, P, l_w, l_i, l_q);
        __syncthreads();
        for (int t = hw; t < PME_AB; t += 8) {                          // both halves of a wave run all 8 rounds (shuffles)
            const T q = l_q[t];
            T sx, sy, sz;
            pme_gather_one<T, ORDER>(t, sub, q, l_w, l_i, grid, P, sx, sy, sz);
            if (sub == 0 && q != T(0)) {
                auto f = frc[a0 + t];
                f.x -= q * sx; f.y -= q * sy; f.z -= q * sz;
                frc[a0 + t] = f;
            }
        }
    }
}

template <class T, int ORDER>
__global__ void __launch_bounds__(256) k_pme_gather(int64_t n_atoms, const typename Vec<T>::T4* __restrict__ pos, const T* __restrict__ grid,
                                                    typename Vec<T>::T4* frc, PmeP<T> P) {
    pme_gather_blocks<T, ORDER>((int)blockIdx.x, (int)gridDim.x, n_atoms, pos, grid, frc, P);
}

// The 3-D transform, one launch per axis, as direct DFTs on LDS-staged line tiles: a block stages C whole lines ([j][c],
// c = line within the tile) and every thread owns ONE output (k, c), so a pass is many short, independent waves.
// The charge mesh is real, so only the half spectrum kz = 0 … nz/2 is carried (Hermitian symmetry halves every pass):
//   k_pme_z_r2c   real lines → nzh = nz/2+1 complex outputs per line (and zeroes the charge mesh behind it)
//   k_pme_dft     y, then x, on the half grid (nx, ny, nzh); the x pass (CONV) does forward · influence function · backward
//   k_pme_z_c2r   Hermitian half lines → real potential mesh, φ[z] = Re X₀ + 2 Σ_{0<k<nz/2} Re(X_k w^{kz}) (+ Nyquist term)
// Half-grid layout: complex<T> at ((x·ny + y)·nzh + kz).  Line q of the y pass = x·nzh + kz (start x·ny·nzh + kz, stride nzh),
// of the x pass = y·nzh + kz (start q, stride ny·nzh).
constexpr int PME_THREADS = 256;

template <class T> struct DftArgs {
    typename Vec<T>::T2* grid;            // half grid
    T* rgrid;                             // r2c: [nx·ny·nz] real charge mesh (read, then zeroed for the next spread)
    T* phi;                               // c2r: [nx·ny·nz] real potential mesh
    const typename Vec<T>::T2* tw;        // [n] e^{-2πi m/n} of this axis
    const T* mh[3];                       // [n_d] signed frequency / L_d
    const T* bsm[3];                      // [n_d] B-spline moduli
    double* e_part;                       // CONV && ENERGY: per-block Σ w·eterm·|S|² over the half spectrum (w = 2 off the symmetry planes)
    PmeP<T> P;
    int axis, sign, C, nzh;               // sign −1: forward (plan_fft!), +1: backward (plan_bfft!), both unnormalised
};

// (the three transform kernels as device bodies: bid = workgroup of the pass, pme_smem = its dynamic LDS — forces_gs.hip runs them beside
// slices of the pair pass in one launch)
template <class T>
__device__ inline void pme_z_r2c_body(const DftArgs<T>& A, int bid, unsigned char* pme_smem) {
    using T2 = typename Vec<T>::T2;
    const int nz = A.P.n[2], nzh = A.nzh, C = A.C, tid = threadIdx.x;
    T2* l_tw = reinterpret_cast<T2*>(pme_smem);
    T* l_a = reinterpret_cast<T*>(l_tw + nz);                          // [j][c] real
    const int64_t n_mesh = (int64_t)A.P.n[0] * A.P.n[1] * nz, n_lines = n_mesh / nz, q0 = (int64_t)bid * C;
    const int n_here = (int)min((int64_t)C, n_lines - q0);
    for (int m = tid; m < nz; m += PME_THREADS) l_tw[m] = A.tw[m];
    for (int e = tid; e < nz * C; e += PME_THREADS) {
        const int c = e / nz, j = e - c * nz;
        T v = T(0);
        if (c < n_here) {
            const int64_t at = (q0 + c) * nz + j;
            v = A.rgrid[at]; A.rgrid[at] = T(0);
        }
        l_a[j * C + c] = v;
    }
    __syncthreads();
    for (int o = tid; o < nzh * C; o += PME_THREADS) {
        const int c = o / nzh, k = o - c * nzh;
        T r0 = T(0), i0 = T(0), r1 = T(0), i1 = T(0);
        int m = 0, j = 0;
        for (; j + 1 < nz; j += 2) {
            const T v0 = l_a[j * C + c]; const T2 w0 = l_tw[m];
            m += k; m -= m >= nz ? nz : 0;
            const T v1 = l_a[(j + 1) * C + c]; const T2 w1 = l_tw[m];
            m += k; m -= m >= nz ? nz : 0;
            r0 += v0 * w0.x; i0 += v0 * w0.y; r1 += v1 * w1.x; i1 += v1 * w1.y;
        }
        if (j < nz) { const T v0 = l_a[j * C + c]; const T2 w0 = l_tw[m]; r0 += v0 * w0.x; i0 += v0 * w0.y; }
        if (c < n_here) { T2 v; v.x = r0 + r1; v.y = i0 + i1; A.grid[(q0 + c) * nzh + k] = v; }
    }
}

template <class T>
__global__ void __launch_bounds__(PME_THREADS) k_pme_z_r2c(DftArgs<T> A) {
    extern __shared__ __align__(16) unsigned char pme_smem[];
    pme_z_r2c_body<T>(A, (int)blockIdx.x, pme_smem);
}

template <class T>
__device__ inline void pme_z_c2r_body(const DftArgs<T>& A, int bid, unsigned char* pme_smem) {
    using T2 = typename Vec<T>::T2;
    const int nz = A.P.n[2], nzh = A.nzh, C = A.C, tid = threadIdx.x;
    T2* l_tw = reinterpret_cast<T2*>(pme_smem);
    T2* l_a = l_tw + nz;                                               // [k][c] complex, k < nzh
    const int64_t n_lines = (int64_t)A.P.n[0] * A.P.n[1], q0 = (int64_t)bid * C;
    const int n_here = (int)min((int64_t)C, n_lines - q0);
    for (int m = tid; m < nz; m += PME_THREADS) { T2 w = A.tw[m]; w.y = -w.y; l_tw[m] = w; }      // e^{+2πi m/n}
    for (int e = tid; e < nzh * C; e += PME_THREADS) {
        const int c = e / nzh, k = e - c * nzh;
        T2 v; v.x = T(0); v.y = T(0);
        if (c < n_here) {
            v = A.grid[(q0 + c) * nzh + k];
            const bool plane = k == 0 || 2 * k == nz;                   // self-conjugate terms count once, the others stand for k and nz − k
            if (!plane) { v.x += v.x; v.y += v.y; }
        }
        l_a[k * C + c] = v;
    }
    __syncthreads();
    for (int o = tid; o < nz * C; o += PME_THREADS) {
        const int c = o / nz, z = o - c * nz;
        T r0 = T(0), r1 = T(0);
        int m = 0, k = 0;
        for (; k + 1 < nzh; k += 2) {
            const T2 v0 = l_a[k * C + c], w0 = l_tw[m];
            m += z; m -= m >= nz ? nz : 0;
            const T2 v1 = l_a[(k + 1) * C + c], w1 = l_tw[m];
            m += z; m -= m >= nz ? nz : 0;
            r0 += v0.x * w0.x - v0.y * w0.y; r1 += v1.x * w1.x - v1.y * w1.y;
        }
        if (k < nzh) { const T2 v0 = l_a[k * C + c], w0 = l_tw[m]; r0 += v0.x * w0.x - v0.y * w0.y; }
        if (c < n_here) A.phi[(q0 + c) * nz + z] = r0 + r1;
    }
}

template <class T>
__global__ void __launch_bounds__(PME_THREADS) k_pme_z_c2r(DftArgs<T> A) {
    extern __shared__ __align__(16) unsigned char pme_smem[];
    pme_z_c2r_body<T>(A, (int)blockIdx.x, pme_smem);
}

// recip_conv_inner! (ewald.jl:678-723) for one wave vector (kx, ky, kz) ≠ 0 of the half spectrum holding S = (re, im): the influence function, and (ENERGY) its
// share of Σ eterm·|S|² and of the six components of the reciprocal virial.  mh[d][k]: signed frequency / L_d (triclinic: the signed integer frequency).
template <class T, bool ENERGY>
__device__ inline T pme_influence(const PmeP<T>& P, const T* const* mh, const T* const* bsm, int kx, int ky, int kz, T re, T im, double& e_loc, double* v_loc) {
    const int nx = P.n[0], ny = P.n[1], nz = P.n[2];
    T mhx = mh[0][kx], mhy = mh[1][ky], mhz = mh[2][kz];
    if (P.tri) {      // m · recip_box (:688-694)
        const T mx = mhx, my = mhy, mz = mhz;
        mhx = mx * P.r[0][0]; mhy = mx * P.r[1][0] + my * P.r[1][1]; mhz = mx * P.r[2][0] + my * P.r[2][1] + mz * P.r[2][2];
    }
    const T m2 = mhx * mhx + mhy * mhy + mhz * mhz;
    const T bprod = (P.pi_V * bsm[0][kx]) * bsm[1][ky] * bsm[2][kz];
    const T denom = m2 * bprod;
    T eterm = P.f_div_er * M<T>::exp(-P.factor * m2) / denom;
    [[maybe_unused]] const T eterm_own = eterm;
    [[maybe_unused]] T wx = -mhx, wy = -mhy, wz = -mhz, eterm_mir = eterm;      // wave vector and influence function of the mirror −k (differ from −m, eterm on Nyquist planes of a sheared cell)
    if (P.tri && (2 * kx == nx || 2 * ky == ny || 2 * kz == nz)) {
        // The reference visits the full mesh and takes the real part of the backward transform.  The frequency of a Nyquist index keeps its
        // sign under the mirror k → −k (:685-693), so on a sheared cell |m|² — and the influence function — of k and of its mirror differ
        // there, the product mesh is not Hermitian, and its real part is the transform of the Hermitian part: S(k) · (eterm(k) + eterm(−k))/2.
        // The half spectrum carries that average (the energy's pair k, −k sums to the same).
        const int jx = kx ? nx - kx : 0, jy = ky ? ny - ky : 0, jz = kz ? nz - kz : 0;
        const T ux = mh[0][jx], uy = mh[1][jy], uz = mh[2][jz];
        const T vx = ux * P.r[0][0], vy = ux * P.r[1][0] + uy * P.r[1][1], vz = ux * P.r[2][0] + uy * P.r[2][1] + uz * P.r[2][2];
        const T n2 = vx * vx + vy * vy + vz * vz;
        eterm_mir = P.f_div_er * M<T>::exp(-P.factor * n2) / (n2 * bprod);
        wx = vx; wy = vy; wz = vz;
        eterm = T(0.5) * (eterm + eterm_mir);
    }
    if constexpr (ENERGY) {
        const bool twice = !(kz == 0 || 2 * kz == nz);           // stands for k and its mirror −k of the full mesh
        if (P.tri) {      // sheared cell: k with its own wave vector, and (if this entry stands for it) −k with its own — no sign bookkeeping
            const double s2 = (double)(re * re + im * im);
            auto add = [&](T ax, T ay, T az, T et) {
                const double Ek = (double)et * s2, a2 = (double)ax * ax + (double)ay * ay + (double)az * az, coeff = 2.0 * (1.0 + (double)P.factor * a2) / a2;
                e_loc += Ek;
                v_loc[0] += Ek * (1.0 - coeff * ax * ax); v_loc[1] += Ek * (1.0 - coeff * ay * ay); v_loc[2] += Ek * (1.0 - coeff * az * az);
                v_loc[3] -= Ek * coeff * ax * ay; v_loc[4] -= Ek * coeff * ax * az; v_loc[5] -= Ek * coeff * ay * az;
            };
            add(mhx, mhy, mhz, eterm_own);
            if (twice) add(wx, wy, wz, eterm_mir);
            return eterm;
        }
        const double Ek = (double)(eterm * (re * re + im * im)) * (twice ? 2.0 : 1.0);
        e_loc += Ek;
        // V·P_k = E_k [I − 2(1 + factor·m²)(m ⊗ m)/m²]  (recip_conv_inner! :701-723), six independent components.
        // The reference visits k and −k separately and an even mesh's Nyquist index keeps its sign of m under the mirror
        // (:685-693), so there the mixed term of a Nyquist axis and a regular one cancels between the two.
        const double coeff = 2.0 * (1.0 + (double)P.factor * (double)m2) / (double)m2;
        const bool nqx = twice && 2 * kx == nx, nqy = twice && 2 * ky == ny;
        v_loc[0] += Ek * (1.0 - coeff * mhx * mhx); v_loc[1] += Ek * (1.0 - coeff * mhy * mhy); v_loc[2] += Ek * (1.0 - coeff * mhz * mhz);
        if (nqx == nqy) v_loc[3] -= Ek * coeff * mhx * mhy;
        if (!nqx) v_loc[4] -= Ek * coeff * mhx * mhz;
        if (!nqy) v_loc[5] -= Ek * coeff * mhy * mhz;
    }
    return eterm;
}

template <class T, bool CONV, bool ENERGY>
__device__ inline void pme_dft_body(const DftArgs<T>& A, int bid, int n_blocks_pass, unsigned char* pme_smem) {
    using T2 = typename Vec<T>::T2;
    const int n = A.P.n[A.axis], nx = A.P.n[0], ny = A.P.n[1], nzh = A.nzh, C = A.C;
    T2* l_tw = reinterpret_cast<T2*>(pme_smem);
    T2* l_a = l_tw + n;
    [[maybe_unused]] T2* l_b = l_a + n * C;
    const int tid = threadIdx.x;
    const int64_t n_lines = (int64_t)nx * ny * nzh / n, q0 = (int64_t)bid * C;
    const int n_here = (int)min((int64_t)C, n_lines - q0);
    const int64_t stride = A.axis == 1 ? nzh : (int64_t)ny * nzh;
    auto line_base = [&](int64_t q) -> int64_t {
        if (A.axis == 1) { const int64_t x = q / nzh; return x * ny * nzh + (q - x * nzh); }
        return q;
    };
    for (int m = tid; m < n; m += PME_THREADS) { T2 w = A.tw[m]; if (A.sign > 0 && !CONV) w.y = -w.y; l_tw[m] = w; }
    for (int e = tid; e < n * C; e += PME_THREADS) {
        const int j = e / C, c = e - j * C;
        T2 v; v.x = T(0); v.y = T(0);
        if (c < n_here) v = A.grid[line_base(q0 + c) + j * stride];
        l_a[j * C + c] = v;
    }
    __syncthreads();
    // output (k, c) = Σ_j src[j][c] · w^(jk), two terms in flight per iteration
    auto dft_one = [&](const T2* src, bool conj, int k, int c, T& re, T& im) {
        T r0 = T(0), i0 = T(0), r1 = T(0), i1 = T(0);
        const T sg = conj ? T(-1) : T(1);
        int m = 0, j = 0;
        for (; j + 1 < n; j += 2) {
            const T2 v0 = src[j * C + c], w0 = l_tw[m];
            m += k; m -= m >= n ? n : 0;
            const T2 v1 = src[(j + 1) * C + c], w1 = l_tw[m];
            m += k; m -= m >= n ? n : 0;
            const T w0y = w0.y * sg, w1y = w1.y * sg;
            r0 += v0.x * w0.x - v0.y * w0y; i0 += v0.x * w0y + v0.y * w0.x;
            r1 += v1.x * w1.x - v1.y * w1y; i1 += v1.x * w1y + v1.y * w1.x;
        }
        if (j < n) {
            const T2 v0 = src[j * C + c], w0 = l_tw[m];
            const T w0y = w0.y * sg;
            r0 += v0.x * w0.x - v0.y * w0y; i0 += v0.x * w0y + v0.y * w0.x;
        }
        re = r0 + r1; im = i0 + i1;
    };
    [[maybe_unused]] double e_loc = 0, v_loc[6] = {0, 0, 0, 0, 0, 0};
    for (int o = tid; o < n * C; o += PME_THREADS) {       // one round unless n > 256
        const int c = o / n, k = o - c * n;
        T re, im;
        dft_one(l_a, false, k, c, re, im);
        if constexpr (!CONV) {
            if (c < n_here) { T2 v; v.x = re; v.y = im; A.grid[line_base(q0 + c) + k * stride] = v; }
        } else {
            T2 v; v.x = T(0); v.y = T(0);
            if (c < n_here) {
                const int kx = k; const int64_t q = q0 + c; const int ky = (int)(q / nzh), kz = (int)(q - (int64_t)ky * nzh);
                if (kx | ky | kz) {
                    const T eterm = pme_influence<T, ENERGY>(A.P, A.mh, A.bsm, kx, ky, kz, re, im, e_loc, v_loc);
                    v.x = re * eterm; v.y = im * eterm;
                }
                // k = 0: the reference leaves the DC term of the charge grid untouched (:681-683); it only adds a constant to the
                // potential mesh, which the B-spline derivative weights (Σ dθ = 0) remove from every force — zero it instead
            }
            l_b[k * C + c] = v;
        }
    }
    if constexpr (CONV) {
        __syncthreads();
        for (int o = tid; o < n * C; o += PME_THREADS) {
            const int c = o / n, k = o - c * n;
            T re, im;
            dft_one(l_b, true, k, c, re, im);
            if (c < n_here) { T2 v; v.x = re; v.y = im; A.grid[line_base(q0 + c) + k * stride] = v; }
        }
        if constexpr (ENERGY) {   // per-block sums, component-major in e_part: energy, then xx yy zz xy xz yz of the virial
            __shared__ double sh_e[PME_THREADS / 64];
            for (int c = 0; c < 7; ++c) {
                double val = c == 0 ? e_loc : v_loc[c - 1];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
                if ((tid & 63) == 0) sh_e[tid >> 6] = val;
                __syncthreads();
                if (tid == 0) { double s = 0; for (int w = 0; w < PME_THREADS / 64; ++w) s += sh_e[w]; A.e_part[(int64_t)c * n_blocks_pass + bid] = s; }
                __syncthreads();
            }
        }
    }
}

template <class T, bool CONV, bool ENERGY>
__global__ void __launch_bounds__(PME_THREADS) k_pme_dft(DftArgs<T> A) {
    extern __shared__ __align__(16) unsigned char pme_smem[];
    pme_dft_body<T, CONV, ENERGY>(A, (int)blockIdx.x, (int)gridDim.x, pme_smem);
}

// the influence function over the half spectrum as a pass of its own (meshes whose transforms run in the FFT library, pme_fft.h): S(k) *= eterm(k), and
// (ENERGY) the per-block sums of the energy and the virial in the layout of k_pme_dft's
template <class T, bool ENERGY>
__global__ void __launch_bounds__(PME_THREADS) k_pme_conv(DftArgs<T> A) {
    using T2 = typename Vec<T>::T2;
    const int nx = A.P.n[0], ny = A.P.n[1], nzh = A.nzh, tid = threadIdx.x;
    const int64_t n = (int64_t)nx * ny * nzh;
    [[maybe_unused]] double e_loc = 0, v_loc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t e = blockIdx.x * (int64_t)PME_THREADS + tid; e < n; e += (int64_t)gridDim.x * PME_THREADS) {
        const int kz = (int)(e % nzh), ky = (int)((e / nzh) % ny), kx = (int)(e / ((int64_t)nzh * ny));
        T2 v = A.grid[e];
        if (kx | ky | kz) {
            const T eterm = pme_influence<T, ENERGY>(A.P, A.mh, A.bsm, kx, ky, kz, v.x, v.y, e_loc, v_loc);
            v.x *= eterm; v.y *= eterm;
        } else { v.x = T(0); v.y = T(0); }      // (k = 0 zeroed: see k_pme_dft)
        A.grid[e] = v;
    }
    if constexpr (ENERGY) {
        __shared__ double sh_e[PME_THREADS / 64];
        for (int c = 0; c < 7; ++c) {
            double val = c == 0 ? e_loc : v_loc[c - 1];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
            if ((tid & 63) == 0) sh_e[tid >> 6] = val;
            __syncthreads();
            if (tid == 0) { double s = 0; for (int w = 0; w < PME_THREADS / 64; ++w) s += sh_e[w]; A.e_part[(int64_t)c * gridDim.x + blockIdx.x] = s; }
            __syncthreads();
        }
    }
}

template <class U> struct PBuf {
    U* p = nullptr; size_t n = 0;
    void set(const std::vector<U>& h) { release(); n = h.size(); if (n) { MHIP_HIP(hipMalloc((void**)&p, n * sizeof(U))); MHIP_HIP(hipMemcpy(p, h.data(), n * sizeof(U), hipMemcpyHostToDevice)); } }
    void alloc(size_t m) { release(); n = m; if (n) MHIP_HIP(hipMalloc((void**)&p, n * sizeof(U))); }
    void update(const std::vector<U>& h) { if (h.size() != n) { set(h); return; } if (n) MHIP_HIP(hipMemcpy(p, h.data(), n * sizeof(U), hipMemcpyHostToDevice)); }      // same length: the allocation stays
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

template <class T> struct Pme {
    using T4 = typename Vec<T>::T4;
    using T2 = typename Vec<T>::T2;
    int order = 0;
    PmeP<T> P;
    PBuf<T2> grid, tw[3];
    PBuf<T> mh[3], bsm[3], rgrid, phi; // rgrid: real charge mesh; phi: real potential mesh
    int nzh = 0;                       // half-spectrum length along z
    bool fft = false; FftPlan3d plan;  // transforms by the FFT library: an axis longer than 512 points (MOLLYHIP_PME_FFT=1: always, =0: never)
    double self_factor = 0, charge_factor = 0;   // E_self = −f/ϵr·α/√π·Σq²  and  E_charge = −f/ϵr·π/(2Vα²)·(Σq)²   (:917-927)

    bool on() const { return order > 0; }
    void release() { plan.destroy(); fft = false; grid.release(); rgrid.release(); phi.release(); for (int d = 0; d < 3; ++d) { tw[d].release(); mh[d].release(); bsm[d].release(); } order = 0; }

    // pme_bspline_moduli (:311-358), in T like the reference
    static void moduli(int ord, const int* n, std::vector<T>* out) {
        const int nmax = std::max(n[0], std::max(n[1], n[2]));
        std::vector<T> data(ord, T(0)), bd(nmax + ord + 1, T(0));
        data[0] = T(1);
        for (int k = 3; k <= ord - 1; ++k) {
            T d = T(1) / (T(k) - T(1));
            data[k - 1] = T(0);
            for (int l = 1; l <= k - 2; ++l) data[k - l - 1] = d * (T(l) * data[k - l - 2] + T(k - l) * data[k - l - 1]);
            data[0] *= d;
        }
        T d = T(1) / (T(ord) - T(1));
        data[ord - 1] = T(0);
        for (int l = 1; l <= ord - 2; ++l) data[ord - l - 1] = d * (T(l) * data[ord - l - 2] + T(ord - l) * data[ord - l - 1]);
        data[0] *= d;
        for (int i = 1; i <= ord; ++i) bd[i] = data[i - 1];
        for (int dd = 0; dd < 3; ++dd) {
            const int nd = n[dd];
            out[dd].assign(nd, T(0));
            for (int i = 1; i <= nd; ++i) {
                T sc = T(0), ss = T(0);
                for (int j = 1; j <= nd; ++j) { T arg = T(2) * T(M_PI) * T(i - 1) * T(j - 1) / T(nd); sc += bd[j - 1] * std::cos(arg); ss += bd[j - 1] * std::sin(arg); }
                out[dd][i - 1] = sc * sc + ss * ss;
            }
            for (int i = 1; i <= nd; ++i) if (out[dd][i - 1] < T(1e-7)) out[dd][i - 1] = (out[dd][(i - 2 + nd) % nd] + out[dd][i % nd]) / T(2);
        }
    }

    // what of the set-up depends on the BOX (as opposed to order, mesh and α): the scaling fields of P, the reciprocal box, the volume in the influence function and in
    // the charge term, and the signed frequencies k / L of an orthorhombic box.  setup() calls it; so does rebox() when mhip_set_box replaces the boundary of a live
    // context — the meshes, twiddle factors, B-spline moduli and library plans stay as they are, nothing is reallocated.
    void box_fields(double alpha, double ke, double eps_r, const double* box, const double* bv9) {
        const T a = T(alpha);
        T V = T(1);
        for (int d = 0; d < 3; ++d) { P.invL[d] = T(1) / T(box[d]); P.n_over_L[d] = T(P.n[d]) * (T(1) / T(box[d])); V *= T(box[d]); }
        P.tri = bv9 ? 1 : 0;
        for (int e = 0; e < 3; ++e) for (int d = 0; d < 3; ++d) P.r[e][d] = T(0);
        if (bv9) {      // invert_box_vectors(::TriclinicBoundary), spatial.jl:338-347, in T
            T bv[3][3]; for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) bv[i][k] = T(bv9[3 * i + k]);
            const T vol = bv[0][0] * bv[1][1] * bv[2][2];
            P.r[0][0] = (bv[1][1] * bv[2][2]) / vol;
            P.r[1][0] = (-bv[1][0] * bv[2][2]) / vol; P.r[1][1] = (bv[0][0] * bv[2][2]) / vol;
            P.r[2][0] = (bv[1][0] * bv[2][1] - bv[1][1] * bv[2][0]) / vol; P.r[2][1] = (-bv[0][0] * bv[2][1]) / vol; P.r[2][2] = (bv[0][0] * bv[1][1]) / vol;
        } else for (int d = 0; d < 3; ++d) P.r[d][d] = P.invL[d];
        P.f_div_er = T(ke) / T(eps_r); P.factor = T(M_PI) * T(M_PI) / (a * a); P.pi_V = T(M_PI) * V;
        self_factor = -(double)P.f_div_er * (double)a / std::sqrt(M_PI);
        charge_factor = -(double)P.f_div_er * M_PI / (2.0 * (double)V * (double)a * (double)a);
        for (int d = 0; d < 3; ++d) {
            const int nd = P.n[d];
            std::vector<T> m(nd);
            const T maxk = T(0.5) * T(nd + 1);
            for (int k = 0; k < nd; ++k) m[k] = (T(k) < maxk ? T(k) : T(k - nd)) * (bv9 ? T(1) : P.invL[d]);      // (triclinic: the signed integer frequency, the kernel multiplies by recip_box)
            mh[d].update(m);
        }
    }
    void rebox(double alpha, double ke, double eps_r, const double* box, const double* bv9) { if (on()) box_fields(alpha, ke, eps_r, box, bv9); }

    // bv9 (nullable): the basis vectors of a TriclinicBoundary, row-major; box = (v1.x, v2.y, v3.z) then (box_sides, spatial.jl:359: mesh sizes and volume come from it)
    void setup(int ord, const int32_t* mesh, double alpha, double ke, double eps_r, const double* box, const int* periodic, const double* bv9 = nullptr) {
        release();
        if (ord == 0) return;
        if (ord != 4 && ord != 5 && ord != 6) throw ApiError{MHIP_ERR_UNSUPPORTED, "PME B-spline order must be 4, 5 or 6"};
        if (!(alpha > 0) || !(eps_r > 0)) throw ApiError{MHIP_ERR_INVALID, "PME needs alpha > 0 and eps_r > 0"};
        for (int d = 0; d < 3; ++d) {
            if (!periodic[d]) throw ApiError{MHIP_ERR_UNSUPPORTED, "PME needs a fully periodic box"};
            // the axis transforms are direct DFTs on LDS-staged lines (one output per thread, any length: the reference's meshes have
            // factors like 23 and 17): O(n) work per mesh point and pass.  That is the right trade for 6mrr-class meshes (46x46x51: each
            // pass sits at the launch floor) and the wrong one for the meshes of 100 nm boxes; past 512 points per axis the call is
            // refused instead of quietly taking tens of milliseconds per step (an FFT library path is not linked).
            if (mesh[d] < ord || mesh[d] > 4096) throw ApiError{MHIP_ERR_UNSUPPORTED, "PME mesh size out of range: B-spline order .. 4096 points per axis"};
        }
        // Past 512 points on an axis the passes above would quietly take tens of milliseconds; such meshes (100 nm boxes) go through the FFT library instead
        // (pme_fft.h: hipFFT real ↔ complex 3-D plans on the same half spectrum), with the influence function as a pass of its own.
        static const int fft_env = [] { const char* v = std::getenv("MOLLYHIP_PME_FFT"); return v && *v ? std::atoi(v) : -1; }();
        const bool long_axis = mesh[0] > 512 || mesh[1] > 512 || mesh[2] > 512;
        if (long_axis && fft_env == 0) throw ApiError{MHIP_ERR_UNSUPPORTED, "PME mesh axis beyond 512 points needs the FFT path (MOLLYHIP_PME_FFT=0 turned it off)"};
        if ((int64_t)mesh[0] * mesh[1] * mesh[2] > ((int64_t)1 << 30)) throw ApiError{MHIP_ERR_CAPACITY, "PME mesh too large"};
        order = ord;
        for (int d = 0; d < 3; ++d) P.n[d] = mesh[d];
        box_fields(alpha, ke, eps_r, box, bv9);
        std::vector<T> bm[3];
        moduli(ord, P.n, bm);
        for (int d = 0; d < 3; ++d) {
            const int nd = P.n[d];
            std::vector<T2> w(nd);
            for (int k = 0; k < nd; ++k) {
                const double ang = -2.0 * M_PI * k / nd;
                w[k].x = (T)std::cos(ang); w[k].y = (T)std::sin(ang);
            }
            tw[d].set(w); bsm[d].set(bm[d]);
        }
        nzh = P.n[2] / 2 + 1;
        try {      // (a mesh whose arrays or library work areas do not fit leaves a context without PME, not one with half of it)
            grid.alloc((size_t)P.n[0] * P.n[1] * nzh);
            rgrid.alloc((size_t)P.n[0] * P.n[1] * P.n[2]);
            MHIP_HIP(hipMemset(rgrid.p, 0, rgrid.n * sizeof(T)));     // from here on k_pme_z_r2c leaves the meshes zeroed behind it
            phi.alloc((size_t)P.n[0] * P.n[1] * P.n[2]);
            fft = long_axis || fft_env == 1;
            if (fft) plan.create(P.n[0], P.n[1], P.n[2], sizeof(T) == 8);
        } catch (...) { release(); throw; }
    }

    DftArgs<T> dft_args(int axis, int sign, int C, double* e_part) const {
        DftArgs<T> A;
        A.grid = grid.p; A.rgrid = rgrid.p; A.phi = phi.p; A.tw = tw[axis].p; for (int d = 0; d < 3; ++d) { A.mh[d] = mh[d].p; A.bsm[d] = bsm[d].p; }
        A.e_part = e_part; A.P = P; A.axis = axis; A.sign = sign; A.C = C; A.nzh = nzh;
        return A;
    }
    // lines per block: one output per thread
    int c_r2c() const { return std::max(1, PME_THREADS / nzh); }
    int c_c2r() const { return std::max(1, PME_THREADS / P.n[2]); }
    int c_xy(int axis) const { return std::max(1, PME_THREADS / P.n[axis]); }
    int conv_blocks() const { return fft ? (int)std::min<int64_t>(cdiv((int64_t)P.n[0] * P.n[1] * nzh, (int64_t)PME_THREADS), 2048) : (int)cdiv((int64_t)P.n[1] * nzh, (int64_t)c_xy(0)); }

    // PME_AB atoms per 256-thread block and round; at most 2048 blocks (each then loops over its atom batches)
    static unsigned atom_blocks(int64_t n) { return (unsigned)std::min<int64_t>(cdiv(n, (int64_t)PME_AB), 2048); }
    template <int ORDER> void spread_t(hipStream_t s, int64_t n, const T4* pos) {
        // 64 atoms per spreading batch (measured: 18.9 us with 64, 18.7 with 32, 29.7 with 128 — not the per-batch chain)
        hipLaunchKernelGGL((k_pme_spread<T, ORDER, 64>), dim3((unsigned)std::min<int64_t>(cdiv(n, (int64_t)64), 2048)), dim3(256), 0, s, n, pos, rgrid.p, P);
    }
    template <int ORDER> void gather_t(hipStream_t s, int64_t n, const T4* pos, T4* frc) { hipLaunchKernelGGL((k_pme_gather<T, ORDER>), dim3(atom_blocks(n)), dim3(256), 0, s, n, pos, (const T*)phi.p, frc, P); }
    template <class K> static void big_lds(K kern, size_t lds) {
        if (lds > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    template <bool CONV, bool EN> void dft_xy(hipStream_t s, int axis, int sign, double* e_part) {
        auto kern = k_pme_dft<T, CONV, EN>;
        const int C = c_xy(axis);
        const size_t lds = (size_t)P.n[axis] * sizeof(T2) * (1 + C * (CONV ? 2 : 1));
        big_lds(kern, lds);
        const int64_t n_lines = (int64_t)P.n[0] * P.n[1] * nzh / P.n[axis];
        hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(n_lines, (int64_t)C)), dim3(PME_THREADS), lds, s, dft_args(axis, sign, C, e_part));
    }
    // geometry of the passes, for launches that run them beside other jobs (forces_gs.hip)
    size_t lds_r2c() const { return (size_t)P.n[2] * sizeof(T2) + (size_t)P.n[2] * c_r2c() * sizeof(T); }
    int blocks_r2c() const { return cdiv((int64_t)P.n[0] * P.n[1], (int64_t)c_r2c()); }
    size_t lds_xy(int axis, bool conv) const { return (size_t)P.n[axis] * sizeof(T2) * (1 + c_xy(axis) * (conv ? 2 : 1)); }
    int blocks_xy(int axis) const { return cdiv((int64_t)P.n[0] * P.n[1] * nzh / P.n[axis], (int64_t)c_xy(axis)); }
    void z_r2c(hipStream_t s) {
        const int C = c_r2c();
        const size_t lds = (size_t)P.n[2] * sizeof(T2) + (size_t)P.n[2] * C * sizeof(T);
        big_lds(k_pme_z_r2c<T>, lds);
        hipLaunchKernelGGL(k_pme_z_r2c<T>, dim3((unsigned)cdiv((int64_t)P.n[0] * P.n[1], (int64_t)C)), dim3(PME_THREADS), lds, s, dft_args(2, -1, C, nullptr));
    }
    void z_c2r(hipStream_t s) {
        const int C = c_c2r();
        const size_t lds = (size_t)P.n[2] * sizeof(T2) + (size_t)nzh * C * sizeof(T2);
        big_lds(k_pme_z_c2r<T>, lds);
        hipLaunchKernelGGL(k_pme_z_c2r<T>, dim3((unsigned)cdiv((int64_t)P.n[0] * P.n[1], (int64_t)C)), dim3(PME_THREADS), lds, s, dft_args(2, +1, C, nullptr));
    }

    // charge mesh → (energy sums, potential mesh): the direct-DFT passes, or the FFT library around the influence-function pass
    void mesh_to_potential(hipStream_t s, double* e_part, bool want_phi) {
        if (fft) {
            plan.forward(s, rgrid.p, grid.p);
            MHIP_HIP(hipMemsetAsync(rgrid.p, 0, rgrid.n * sizeof(T), s));      // (the passes leave the charge mesh zeroed for the next spreading; so does this path)
            DftArgs<T> A = dft_args(0, -1, 1, e_part);
            const int nb = conv_blocks();
            if (e_part) hipLaunchKernelGGL((k_pme_conv<T, true>), dim3(nb), dim3(PME_THREADS), 0, s, A);
            else hipLaunchKernelGGL((k_pme_conv<T, false>), dim3(nb), dim3(PME_THREADS), 0, s, A);
            if (want_phi) plan.backward(s, grid.p, phi.p);
            return;
        }
        z_r2c(s);
        dft_xy<false, false>(s, 1, -1, nullptr);
        if (e_part) dft_xy<true, true>(s, 0, -1, e_part); else dft_xy<true, false>(s, 0, -1, nullptr);
        if (want_phi) { dft_xy<false, false>(s, 1, +1, nullptr); z_c2r(s); }
    }

    // ewald_pe_forces! (:873-929) on the sorted arrays: frc (nullable) gets the reciprocal-space forces ADDED; e_part (nullable)
    // receives 7·conv_blocks() partial sums, component-major: Σ eterm·|S|² and the six components of the reciprocal virial (the caller halves
    // them and adds the self / net-charge terms).
    void run(hipStream_t s, int64_t n_atoms, const T4* pos, T4* frc, double* e_part) {
        if (order == 4) spread_t<4>(s, n_atoms, pos); else if (order == 5) spread_t<5>(s, n_atoms, pos); else spread_t<6>(s, n_atoms, pos);
        mesh_to_potential(s, e_part, frc != nullptr);
        if (frc) {
            if (order == 4) gather_t<4>(s, n_atoms, pos, frc); else if (order == 5) gather_t<5>(s, n_atoms, pos, frc); else gather_t<6>(s, n_atoms, pos, frc);
        }
        MHIP_HIP(hipGetLastError());
    }
};

}  // namespace mhip

// bonded.h — specific (bonded) interactions that ride along with the pairwise path for a solvated protein:
// HarmonicBond, HarmonicAngle, PeriodicTorsion and EwaldExclusion.  One thread per term, hardware float
// atomics into the sorted force array (≙ specific_force_{2,3,4}_atoms_kernel!, src/kernels.jl:233-342).
// Behavioural spec: harmonic_bond.jl:44-54, harmonic_angle.jl:46-67, periodic_torsion.jl:93-142,
// spatial.jl:834-894, ewald.jl:1019-1055.
#pragma once
#include <cstdlib>
#include <vector>

#include "physics.h"

namespace mhip {

template <class T> __device__ inline void min_image(const typename Vec<T>::T4& a, const typename Vec<T>::T4& b, const GridP<T>& G, T* d) {
    d[0] = G.periodic[0] ? vector_1d_exact(a.x, b.x, G.L[0]) : b.x - a.x;
    d[1] = G.periodic[1] ? vector_1d_exact(a.y, b.y, G.L[1]) : b.y - a.y;
    d[2] = G.periodic[2] ? vector_1d_exact(a.z, b.z, G.L[2]) : b.z - a.z;
}
template <class T> __device__ inline void cross3(const T* a, const T* b, T* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
template <class T> __device__ inline T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> __device__ inline void add_force(typename Vec<T>::T4* frc, int s, T fx, T fy, T fz) {
    atomicAdd(&frc[s].x, fx); atomicAdd(&frc[s].y, fy); atomicAdd(&frc[s].z, fz);
}
// block-level sum of one double per thread into part[blockIdx.x] (256 threads)
__device__ inline void block_sum_to(double v, double* part) {
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double a = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) a += sh[q]; part[blockIdx.x] = a; }
}

template <class T, bool ENERGY>
__device__ inline void d_bonds(int64_t t, int64_t n, const int32_t* __restrict__ bi, const int32_t* __restrict__ bj, const T* __restrict__ bk, const T* __restrict__ br0,
                        const int32_t* __restrict__ inv, const typename Vec<T>::T4* __restrict__ pos, typename Vec<T>::T4* frc, double& e, const GridP<T>& G) {
    if (t < n) {
        int i = inv[bi[t]], j = inv[bj[t]];
        T ab[3]; min_image<T>(pos[i], pos[j], G, ab);
        T r = M<T>::sqrt(dot3(ab, ab));
        T dr = r - br0[t];
        if constexpr (ENERGY) e = (double)((bk[t] / T(2)) * dr * dr);
        else { T c = bk[t] * dr / r; add_force<T>(frc, i, c * ab[0], c * ab[1], c * ab[2]); add_force<T>(frc, j, -c * ab[0], -c * ab[1], -c * ab[2]); }
    }
}

template <class T, bool ENERGY>
__device__ inline void d_angles(int64_t t, int64_t n, const int32_t* __restrict__ ai, const int32_t* __restrict__ aj, const int32_t* __restrict__ ak, const T* __restrict__ kth,
                         const T* __restrict__ th0, const int32_t* __restrict__ inv, const typename Vec<T>::T4* __restrict__ pos, typename Vec<T>::T4* frc,
                         double& e, const GridP<T>& G) {
    if (t < n) {
        int i = inv[ai[t]], j = inv[aj[t]], k = inv[ak[t]];
        T ba[3], bc[3], cr[3];
        min_image<T>(pos[j], pos[i], G, ba); min_image<T>(pos[j], pos[k], G, bc);
        T nba = M<T>::sqrt(dot3(ba, ba)), nbc = M<T>::sqrt(dot3(bc, bc));
        T cs = dot3(ba, bc) / (nba * nbc);
        cs = cs < T(-1) ? T(-1) : (cs > T(1) ? T(1) : cs);   // acos_bound
        T dth = acos(cs) - th0[t];
        if constexpr (ENERGY) e = (double)((kth[t] / T(2)) * dth * dth);
        else {
            cross3(ba, bc, cr);
            if (!(cr[0] == T(0) && cr[1] == T(0) && cr[2] == T(0))) {
                T pa[3], pc[3], nb[3] = {-bc[0], -bc[1], -bc[2]};
                cross3(ba, cr, pa); cross3(nb, cr, pc);
                T term = -kth[t] * dth;
                T sa = term / (nba * M<T>::sqrt(dot3(pa, pa))), sc = term / (nbc * M<T>::sqrt(dot3(pc, pc)));
                T fa[3] = {sa * pa[0], sa * pa[1], sa * pa[2]}, fc[3] = {sc * pc[0], sc * pc[1], sc * pc[2]};
                add_force<T>(frc, i, fa[0], fa[1], fa[2]);
                add_force<T>(frc, j, -fa[0] - fc[0], -fa[1] - fc[1], -fa[2] - fc[2]);
                add_force<T>(frc, k, fc[0], fc[1], fc[2]);
            }
        }
    }
}

template <class T, bool ENERGY>
__device__ inline void d_torsions(int64_t t, int64_t n, const int32_t* __restrict__ ti, const int32_t* __restrict__ tj, const int32_t* __restrict__ tk, const int32_t* __restrict__ tl,
                           const int32_t* __restrict__ per, const T* __restrict__ phase, const T* __restrict__ k0, const int32_t* __restrict__ inv,
                           const typename Vec<T>::T4* __restrict__ pos, typename Vec<T>::T4* frc, double& e, const GridP<T>& G) {
    if (t < n) {
        int i = inv[ti[t]], j = inv[tj[t]], k = inv[tk[t]], l = inv[tl[t]];
        T ab[3], bc[3], cd[3], c1[3], c2[3], c12[3];
        min_image<T>(pos[i], pos[j], G, ab); min_image<T>(pos[j], pos[k], G, bc); min_image<T>(pos[k], pos[l], G, cd);
        cross3(ab, bc, c1); cross3(bc, cd, c2); cross3(c1, c2, c12);
        T bcn = M<T>::sqrt(dot3(bc, bc));
        T th = atan2(dot3(c12, bc) / bcn, dot3(c1, c2));
        T p = T(per[t]);
        if constexpr (ENERGY) e = (double)(k0[t] + k0[t] * cos(p * th - phase[t]));
        else {
            T dE = -k0[t] * p * sin(p * th - phase[t]);
            T d11 = dot3(c1, c1), d22 = dot3(c2, c2);
            T ca = -dot3(ab, bc) / (bcn * bcn), cb = -dot3(cd, bc) / (bcn * bcn);
            T fi[3], fl[3];
            for (int d = 0; d < 3; ++d) { fi[d] = dE * bcn * c1[d] / d11; fl[d] = -dE * bcn * c2[d] / d22; }
            T v[3] = {ca * fi[0] - cb * fl[0], ca * fi[1] - cb * fl[1], ca * fi[2] - cb * fl[2]};
            add_force<T>(frc, i, fi[0], fi[1], fi[2]);
            add_force<T>(frc, j, v[0] - fi[0], v[1] - fi[1], v[2] - fi[2]);
            add_force<T>(frc, k, -v[0] - fl[0], -v[1] - fl[1], -v[2] - fl[2]);
            add_force<T>(frc, l, fl[0], fl[1], fl[2]);
        }
    }
}

template <class T, bool ENERGY>
__device__ inline void d_ewald_excl(int64_t t, int64_t n, const int32_t* __restrict__ xi, const int32_t* __restrict__ xj, const int32_t* __restrict__ inv,
                             const typename Vec<T>::T4* __restrict__ pos, typename Vec<T>::T4* frc, double& e, const GridP<T>& G, const InterP<T>& I) {
    if (t < n) {
        int i = inv[xi[t]], j = inv[xj[t]];
        auto pi = pos[i], pj = pos[j];
        T d[3]; min_image<T>(pi, pj, G, d);
        T r = M<T>::sqrt(dot3(d, d));
        T ar = I.alpha * r;
        T er = M<T>::erf(ar);
        T kqq = I.ke * (pi.w * pj.w);
        if constexpr (ENERGY) e = (double)(er > T(1e-6) ? -kqq / r * er : -I.alpha * T(2) * kqq / M<T>::sqrt(T(M_PI)));
        else if (er > T(1e-6)) {
            T inv_r = T(1) / r;
            T dE = kqq * inv_r * inv_r * inv_r * (er - I.two_over_sqrt_pi * ar * exp(-(ar * ar)));
            add_force<T>(frc, i, dE * d[0], dE * d[1], dE * d[2]);
            add_force<T>(frc, j, -dE * d[0], -dE * d[1], -dE * d[2]);
        }
    }
}

// All specific interaction lists in ONE launch: consecutive block ranges serve bonds, angles, torsion terms and Ewald
// exclusions, so the (latency-bound, few-thousand-thread) lists overlap instead of queueing behind each other.
template <class T> struct BondedArgs {
    int64_t n_b, n_a, n_t, n_x;
    int blk_b, blk_a, blk_t;          // first block of the angle / torsion / exclusion ranges are the running sums
    const int32_t *b_i, *b_j; const T *b_k, *b_r0;
    const int32_t *a_i, *a_j, *a_k; const T *a_kth, *a_th0;
    const int32_t *t_i, *t_j, *t_k, *t_l, *t_per; const T *t_phase, *t_k0;
    const int32_t *x_i, *x_j;
    const int32_t* inv; const typename Vec<T>::T4* pos; typename Vec<T>::T4* frc; double* part;
    GridP<T> G; InterP<T> I;
};

template <class T, bool ENERGY>
__global__ void k_bonded(BondedArgs<T> A) {
    const int blk = blockIdx.x;
    double e = 0;
    if (blk < A.blk_b) d_bonds<T, ENERGY>((int64_t)blk * blockDim.x + threadIdx.x, A.n_b, A.b_i, A.b_j, A.b_k, A.b_r0, A.inv, A.pos, A.frc, e, A.G);
    else if (blk < A.blk_b + A.blk_a) d_angles<T, ENERGY>((int64_t)(blk - A.blk_b) * blockDim.x + threadIdx.x, A.n_a, A.a_i, A.a_j, A.a_k, A.a_kth, A.a_th0, A.inv, A.pos, A.frc, e, A.G);
    else if (blk < A.blk_b + A.blk_a + A.blk_t) d_torsions<T, ENERGY>((int64_t)(blk - A.blk_b - A.blk_a) * blockDim.x + threadIdx.x, A.n_t, A.t_i, A.t_j, A.t_k, A.t_l, A.t_per, A.t_phase, A.t_k0, A.inv, A.pos, A.frc, e, A.G);
    else d_ewald_excl<T, ENERGY>((int64_t)(blk - A.blk_b - A.blk_a - A.blk_t) * blockDim.x + threadIdx.x, A.n_x, A.x_i, A.x_j, A.inv, A.pos, A.frc, e, A.G, A.I);
    if constexpr (ENERGY) block_sum_to(e, A.part);
}

template <class U> struct HBuf {   // device array filled from a host array once
    U* p = nullptr; size_t n = 0;
    void set(const U* h, size_t m) {
        if (p) (void)hipFree(p);
        p = nullptr; n = m;
        if (m) { MHIP_HIP(hipMalloc((void**)&p, m * sizeof(U))); MHIP_HIP(hipMemcpy(p, h, m * sizeof(U), hipMemcpyHostToDevice)); }
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

template <class T> struct Bonded {
    using T4 = typename Vec<T>::T4;
    HBuf<int32_t> b_i, b_j, a_i, a_j, a_k, t_i, t_j, t_k, t_l, t_per, x_i, x_j;
    HBuf<T> b_k, b_r0, a_kth, a_th0, t_phase, t_k0;

    static void check(int64_t cap, int64_t n, std::initializer_list<const int32_t*> idx) {
        if (n < 0) throw ApiError{MHIP_ERR_INVALID, "negative interaction count"};
        for (const int32_t* a : idx) { if (n && !a) throw ApiError{MHIP_ERR_INVALID, "null index array"}; for (int64_t k = 0; k < n; ++k) if (a[k] < 0 || a[k] >= cap) throw ApiError{MHIP_ERR_INVALID, "specific interaction index out of range"}; }
    }
    void set_bonds(int64_t cap, int64_t n, const int32_t* i, const int32_t* j, const T* k, const T* r0) { check(cap, n, {i, j}); b_i.set(i, n); b_j.set(j, n); b_k.set(k, n); b_r0.set(r0, n); }
    void set_angles(int64_t cap, int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const T* kth, const T* th0) {
        check(cap, n, {i, j, k}); a_i.set(i, n); a_j.set(j, n); a_k.set(k, n); a_kth.set(kth, n); a_th0.set(th0, n);
    }
    void set_torsions(int64_t cap, int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const int32_t* l, const int32_t* per, const T* ph, const T* k0) {
        check(cap, n, {i, j, k, l}); t_i.set(i, n); t_j.set(j, n); t_k.set(k, n); t_l.set(l, n); t_per.set(per, n); t_phase.set(ph, n); t_k0.set(k0, n);
    }
    void set_ewx(int64_t cap, int64_t n, const int32_t* i, const int32_t* j) { check(cap, n, {i, j}); x_i.set(i, n); x_j.set(j, n); }
    void on_reorder() {}   // terms address atoms through inv[]: nothing to rebuild after a re-sort
    bool any() const { return b_i.n || a_i.n || t_i.n || x_i.n; }
    void release() { for (auto* h : {&b_i, &b_j, &a_i, &a_j, &a_k, &t_i, &t_j, &t_k, &t_l, &t_per, &x_i, &x_j}) h->release(); for (auto* h : {&b_k, &b_r0, &a_kth, &a_th0, &t_phase, &t_k0}) h->release(); }

    static constexpr int BT = 64;   // one wave per block: thousands of short, latency-bound terms spread over all CUs
    BondedArgs<T> args(const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv, T4* frc, double* part) const {
        BondedArgs<T> A;
        A.n_b = b_i.n; A.n_a = a_i.n; A.n_t = t_i.n; A.n_x = x_i.n;
        A.blk_b = cdiv(b_i.n, BT); A.blk_a = cdiv(a_i.n, BT); A.blk_t = cdiv(t_i.n, BT);
        A.b_i = b_i.p; A.b_j = b_j.p; A.b_k = b_k.p; A.b_r0 = b_r0.p;
        A.a_i = a_i.p; A.a_j = a_j.p; A.a_k = a_k.p; A.a_kth = a_kth.p; A.a_th0 = a_th0.p;
        A.t_i = t_i.p; A.t_j = t_j.p; A.t_k = t_k.p; A.t_l = t_l.p; A.t_per = t_per.p; A.t_phase = t_phase.p; A.t_k0 = t_k0.p;
        A.x_i = x_i.p; A.x_j = x_j.p; A.inv = inv; A.pos = pos; A.frc = frc; A.part = part; A.G = G; A.I = I;
        return A;
    }
    int n_blocks() const { return cdiv(b_i.n, BT) + cdiv(a_i.n, BT) + cdiv(t_i.n, BT) + cdiv(x_i.n, BT); }

    void launch_forces(hipStream_t s, const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv, T4* frc) {
        int nb = n_blocks();
        if (!nb) return;
        static const bool split = [] { const char* v = std::getenv("MOLLYHIP_BONDED_SPLIT"); return v && *v && std::atoi(v) != 0; }();
        if (split) {   // timing experiments: one launch per interaction type (blocks of the other types see empty lists)
            for (int k = 0; k < 4; ++k) {
                BondedArgs<T> A = args(G, I, pos, inv, frc, nullptr);
                if (k != 0) { A.n_b = 0; A.blk_b = 0; }
                if (k != 1) { A.n_a = 0; A.blk_a = 0; }
                if (k != 2) { A.n_t = 0; A.blk_t = 0; }
                if (k != 3) A.n_x = 0;
                const int g = k == 0 ? A.blk_b : k == 1 ? A.blk_a : k == 2 ? A.blk_t : cdiv(A.n_x, (int64_t)BT);
                if (g > 0) hipLaunchKernelGGL((k_bonded<T, false>), dim3(g), dim3(BT), 0, s, A);
            }
            MHIP_HIP(hipGetLastError());
            return;
        }
        hipLaunchKernelGGL((k_bonded<T, false>), dim3(nb), dim3(BT), 0, s, args(G, I, pos, inv, frc, nullptr));
        MHIP_HIP(hipGetLastError());
    }
    // writes per-block partial energies into part (grown as needed); returns their count
    template <class DB> int launch_energy(hipStream_t s, const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv, DB& part) {
        int nb = n_blocks();
        if (!nb) return 0;
        part.reserve(nb);
        hipLaunchKernelGGL((k_bonded<T, true>), dim3(nb), dim3(BT), 0, s, args(G, I, pos, inv, (T4*)nullptr, part.p));
        MHIP_HIP(hipGetLastError());
        return nb;
    }
};

}  // namespace mhip

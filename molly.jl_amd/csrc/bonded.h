// bonded.h — specific (bonded) interactions that ride along with the pairwise path for a solvated protein:
// HarmonicBond, HarmonicAngle, PeriodicTorsion and EwaldExclusion (≙ specific_force_{2,3,4}_atoms_kernel!, src/kernels.jl:233-342).
// Forces in two atomic-free launches: one thread per term writes the term's 2-4 force vectors into per-term slots, then one thread
// per atom sums the slots of the terms it takes part in (a CSR of slot indices built once).  The one-launch scatter with float
// atomics was bound by atomic throughput (27 us for 0.3 M atomics on 6mrr, 5 us with the atomics compiled out) and its sums were
// not reproducible; energies keep the simple per-term kernel.
// Behavioural spec: harmonic_bond.jl:44-54, harmonic_angle.jl:46-67, periodic_torsion.jl:93-142,
// spatial.jl:834-894, ewald.jl:1019-1055.
#pragma once
#include <cstdlib>
#include <vector>

#include "physics.h"

namespace mhip {

template <class T> __device__ inline void min_image(const typename Vec<T>::T4& a, const typename Vec<T>::T4& b, const GridP<T>& G, T* d) {
    min_image_exact<T>(a.x, a.y, a.z, b.x, b.y, b.z, G, d[0], d[1], d[2]);
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

// Every term hands its per-atom forces to `sink(sorted atom, role in the term, fx, fy, fz)` — on EVERY path, so that a slot sink
// never leaves a stale slot behind.
template <class T> struct AtomicSink {
    typename Vec<T>::T4* frc;
    __device__ inline void operator()(int s, int, T fx, T fy, T fz) const { add_force<T>(frc, s, fx, fy, fz); }
};
template <class T> struct SlotSink {     // slot = first slot of the term + role
    typename Vec<T>::T4* out; int64_t first;
    __device__ inline void operator()(int, int role, T fx, T fy, T fz) const { out[first + role] = make4<T>(fx, fy, fz, T(0)); }
};

// Σ over the atoms of a term of (r_atom − r_first) ⊗ f_atom.  The reference takes the SECOND atom as the origin (force.jl:991-1060);
// a term's forces add up to zero, so the tensor does not depend on the origin (to rounding).
template <class T> struct VirialSink {
    const typename Vec<T>::T4* pos; const GridP<T>* G; double* v; typename Vec<T>::T4 p0;
    __device__ inline void operator()(int s, int role, T fx, T fy, T fz) {
        if (role == 0) { p0 = pos[s]; return; }
        T d[3]; min_image<T>(p0, pos[s], *G, d);
        const T f[3] = {fx, fy, fz};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) v[3 * a + b] += (double)(d[a] * f[b]);
    }
};

template <class T, bool ENERGY, class Sink>
__device__ inline void d_bonds(int64_t t, int64_t n, const int32_t* __restrict__ bi, const int32_t* __restrict__ bj, const T* __restrict__ bk, const T* __restrict__ br0,
                        const int32_t* __restrict__ inv, const typename Vec<T>::T4* __restrict__ pos, Sink&& sink, double& e, const GridP<T>& G) {
    if (t < n) {
        int i = inv[bi[t]], j = inv[bj[t]];
        T ab[3]; min_image<T>(pos[i], pos[j], G, ab);
        T r = M<T>::sqrt(dot3(ab, ab));
        T dr = r - br0[t];
        if constexpr (ENERGY) e = (double)((bk[t] / T(2)) * dr * dr);
        else { T c = bk[t] * dr / r; sink(i, 0, c * ab[0], c * ab[1], c * ab[2]); sink(j, 1, -c * ab[0], -c * ab[1], -c * ab[2]); }
    }
}

template <class T, bool ENERGY, class Sink>
__device__ inline void d_angles(int64_t t, int64_t n, const int32_t* __restrict__ ai, const int32_t* __restrict__ aj, const int32_t* __restrict__ ak, const T* __restrict__ kth,
                         const T* __restrict__ th0, const int32_t* __restrict__ inv, const typename Vec<T>::T4* __restrict__ pos, Sink&& sink,
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
                sink(i, 0, fa[0], fa[1], fa[2]);
                sink(j, 1, -fa[0] - fc[0], -fa[1] - fc[1], -fa[2] - fc[2]);
                sink(k, 2, fc[0], fc[1], fc[2]);
            } else { sink(i, 0, T(0), T(0), T(0)); sink(j, 1, T(0), T(0), T(0)); sink(k, 2, T(0), T(0), T(0)); }   // collinear: no force (harmonic_angle.jl:52-54)
        }
    }
}

template <class T, bool ENERGY, class Sink>
__device__ inline void d_torsions(int64_t t, int64_t n, const int32_t* __restrict__ ti, const int32_t* __restrict__ tj, const int32_t* __restrict__ tk, const int32_t* __restrict__ tl,
                           const int32_t* __restrict__ per, const T* __restrict__ phase, const T* __restrict__ k0, const int32_t* __restrict__ inv,
                           const typename Vec<T>::T4* __restrict__ pos, Sink&& sink, double& e, const GridP<T>& G) {
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
            sink(i, 0, fi[0], fi[1], fi[2]);
            sink(j, 1, v[0] - fi[0], v[1] - fi[1], v[2] - fi[2]);
            sink(k, 2, -v[0] - fl[0], -v[1] - fl[1], -v[2] - fl[2]);
            sink(l, 3, fl[0], fl[1], fl[2]);
        }
    }
}

template <class T, bool ENERGY, class Sink>
__device__ inline void d_ewald_excl(int64_t t, int64_t n, const int32_t* __restrict__ xi, const int32_t* __restrict__ xj, const int32_t* __restrict__ inv,
                             const typename Vec<T>::T4* __restrict__ pos, Sink&& sink, double& e, const GridP<T>& G, const InterP<T>& I) {
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
            sink(i, 0, dE * d[0], dE * d[1], dE * d[2]);
            sink(j, 1, -dE * d[0], -dE * d[1], -dE * d[2]);
        } else { sink(i, 0, T(0), T(0), T(0)); sink(j, 1, T(0), T(0), T(0)); }
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
    int64_t slot_base[4];            // SLOTS: first slot of each type's range (bonds 2, angles 3, torsions 4, exclusions 2 per term)
    GridP<T> G; InterP<T> I;
};

// SLOTS: the forces go to per-term slots (A.frc is then the slot array, type ranges at slot_base[]) instead of atomics.
// (blk, lane, bt) = term block, lane in it, lanes per term block: a kernel of its own uses its grid; the fused launch of
// step_fused.h runs four 64-lane term blocks in each of its 256-lane workgroups.
template <class T, bool ENERGY, bool SLOTS>
__device__ inline void bonded_terms(const BondedArgs<T>& A, int blk, int lane, int bt, double& e) {
    auto run = [&](auto&& mk) {
        if (blk < A.blk_b) { const int64_t t = (int64_t)blk * bt + lane; d_bonds<T, ENERGY>(t, A.n_b, A.b_i, A.b_j, A.b_k, A.b_r0, A.inv, A.pos, mk(A.slot_base[0] + 2 * t), e, A.G); }
        else if (blk < A.blk_b + A.blk_a) { const int64_t t = (int64_t)(blk - A.blk_b) * bt + lane; d_angles<T, ENERGY>(t, A.n_a, A.a_i, A.a_j, A.a_k, A.a_kth, A.a_th0, A.inv, A.pos, mk(A.slot_base[1] + 3 * t), e, A.G); }
        else if (blk < A.blk_b + A.blk_a + A.blk_t) { const int64_t t = (int64_t)(blk - A.blk_b - A.blk_a) * bt + lane; d_torsions<T, ENERGY>(t, A.n_t, A.t_i, A.t_j, A.t_k, A.t_l, A.t_per, A.t_phase, A.t_k0, A.inv, A.pos, mk(A.slot_base[2] + 4 * t), e, A.G); }
        else { const int64_t t = (int64_t)(blk - A.blk_b - A.blk_a - A.blk_t) * bt + lane; d_ewald_excl<T, ENERGY>(t, A.n_x, A.x_i, A.x_j, A.inv, A.pos, mk(A.slot_base[3] + 2 * t), e, A.G, A.I); }
    };
    if constexpr (SLOTS) run([&](int64_t first) { return SlotSink<T>{A.frc, first}; });
    else run([&](int64_t) { return AtomicSink<T>{A.frc}; });
}
template <class T, bool ENERGY, bool SLOTS>
__global__ void k_bonded(BondedArgs<T> A) {
    double e = 0;
    bonded_terms<T, ENERGY, SLOTS>(A, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, e);
    if constexpr (ENERGY) block_sum_to(e, A.part);
}

// the specific interactions' virial: per-block partial sums of the nine components, component-major in A.part
template <class T>
__global__ void k_bonded_virial(BondedArgs<T> A, int n_blocks_total) {
    const int blk = blockIdx.x;
    double e = 0, v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    VirialSink<T> sink{A.pos, &A.G, v, make4<T>(T(0), T(0), T(0), T(0))};
    if (blk < A.blk_b) d_bonds<T, false>((int64_t)blk * blockDim.x + threadIdx.x, A.n_b, A.b_i, A.b_j, A.b_k, A.b_r0, A.inv, A.pos, sink, e, A.G);
    else if (blk < A.blk_b + A.blk_a) d_angles<T, false>((int64_t)(blk - A.blk_b) * blockDim.x + threadIdx.x, A.n_a, A.a_i, A.a_j, A.a_k, A.a_kth, A.a_th0, A.inv, A.pos, sink, e, A.G);
    else if (blk < A.blk_b + A.blk_a + A.blk_t) d_torsions<T, false>((int64_t)(blk - A.blk_b - A.blk_a) * blockDim.x + threadIdx.x, A.n_t, A.t_i, A.t_j, A.t_k, A.t_l, A.t_per, A.t_phase, A.t_k0, A.inv, A.pos, sink, e, A.G);
    else d_ewald_excl<T, false>((int64_t)(blk - A.blk_b - A.blk_a - A.blk_t) * blockDim.x + threadIdx.x, A.n_x, A.x_i, A.x_j, A.inv, A.pos, sink, e, A.G, A.I);
    for (int c = 0; c < 9; ++c) { block_sum_to(v[c], A.part + (int64_t)c * n_blocks_total); __syncthreads(); }
}

// frc[s] += Σ slots of the terms atom orig[s] takes part in.  Eight lanes share an atom (lane l takes slots l, l+8, … and a fixed
// shuffle tree adds them): a backbone atom sits in dozens of torsion terms, and one lane walking them serially set the kernel's
// duration.  Fixed order → bit-reproducible.
constexpr int COLLECT_LANES = 8;
// gt = global lane number (COLLECT_LANES per atom).  ASSIGN: out[s] = the sum (zero for atoms without terms) instead of out[s] += it —
// for a launch that runs next to another writer of the force array (step_fused.h) and leaves its share in a side array.
// parts (nullable): n_parts more force arrays, part_stride atoms apart, whose entry of the atom is added in fixed order — the partial pair
// forces of the group-split pass (forces_gs.hip), folded in by the one launch that visits every atom anyway.
// the slot sums of atom gt / COLLECT_LANES by lane gt % COLLECT_LANES of its group (all COLLECT_LANES lanes of a group call it): the sum — slots in role order,
// then the group-split pass's partial forces — in the group's lane 0; returns whether this lane is that lane of a live atom
template <class T>
__device__ inline bool bonded_collect_sum(int64_t gt, int64_t n_owned, const int32_t* __restrict__ orig, const int32_t* __restrict__ role_start, const int32_t* __restrict__ role_slot,
                                          const typename Vec<T>::T4* __restrict__ slots, const typename Vec<T>::T4* __restrict__ parts, int n_parts, int64_t part_stride,
                                          T& fx, T& fy, T& fz, bool& any) {
    const int64_t s = gt / COLLECT_LANES;
    const int l = (int)(gt % COLLECT_LANES);
    const bool live = s < n_owned;
    int r0 = 0, r1 = 0;
    if (live) { const int a = orig[s]; r0 = role_start[a]; r1 = role_start[a + 1]; }
    fx = T(0); fy = T(0); fz = T(0);
    // four of a lane's slots per round, their (dependent) fetches issued together — index, then record: an atom of the protein interior has 40 slots, five per
    // lane, and one after the other they were ten memory latencies in the one launch of the step that every atom waits for.  Same order of the adds as before.
    for (int r = r0 + l; r < r1; r += 4 * COLLECT_LANES) {
        int ix[4]; typename Vec<T>::T4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ix[k] = role_slot[min(r + k * COLLECT_LANES, r1 - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = slots[ix[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (r + k * COLLECT_LANES < r1) { fx += v[k].x; fy += v[k].y; fz += v[k].z; }
    }
#pragma unroll
    for (int o = COLLECT_LANES / 2; o > 0; o >>= 1) { fx += __shfl_xor(fx, o, 64); fy += __shfl_xor(fy, o, 64); fz += __shfl_xor(fz, o, 64); }
    if (parts && live && l == 0) {      // (the group-split pass's partial forces: requested together, added in order)
        for (int q0 = 0; q0 < n_parts; q0 += 4) {
            typename Vec<T>::T4 pv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = parts[(int64_t)min(q0 + k, n_parts - 1) * part_stride + s];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (q0 + k < n_parts) { fx += pv[k].x; fy += pv[k].y; fz += pv[k].z; }
        }
    }
    any = r1 > r0 || parts;
    return live && l == 0;
}
template <class T, bool ASSIGN>
__device__ inline void bonded_collect_lane(int64_t gt, int64_t n_owned, const int32_t* __restrict__ orig, const int32_t* __restrict__ role_start, const int32_t* __restrict__ role_slot,
                                           const typename Vec<T>::T4* __restrict__ slots, typename Vec<T>::T4* out,
                                           const typename Vec<T>::T4* __restrict__ parts = nullptr, int n_parts = 0, int64_t part_stride = 0) {
    T fx, fy, fz; bool any;
    const bool mine = bonded_collect_sum<T>(gt, n_owned, orig, role_start, role_slot, slots, parts, n_parts, part_stride, fx, fy, fz, any);
    const int64_t s = gt / COLLECT_LANES;
    if constexpr (ASSIGN) {
        if (mine) out[s] = make4<T>(fx, fy, fz, T(0));
    } else if (mine && any) {
        auto f = out[s];
        f.x += fx; f.y += fy; f.z += fz;
        out[s] = f;
    }
}
template <class T>
__global__ void k_bonded_collect(int64_t n_owned, const int32_t* __restrict__ orig, const int32_t* __restrict__ role_start, const int32_t* __restrict__ role_slot,
                                 const typename Vec<T>::T4* __restrict__ slots, typename Vec<T>::T4* frc,
                                 const typename Vec<T>::T4* __restrict__ parts, int n_parts, int64_t part_stride) {
    bonded_collect_lane<T, false>(blockIdx.x * (int64_t)blockDim.x + threadIdx.x, n_owned, orig, role_start, role_slot, slots, frc, parts, n_parts, part_stride);
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
    // slot path: per-atom CSR of slot indices (rebuilt lazily after any set_*), slot array
    std::vector<int32_t> h_idx[4][4]; HBuf<int32_t> role_start, role_slot; bool roles_dirty = true; int64_t roles_cap = 0, n_slots = 0, slot_base[4] = {0, 0, 0, 0};
    T4* slots = nullptr; size_t slots_cap = 0;
    void keep(int type, int64_t n, std::initializer_list<const int32_t*> idx) {
        int r = 0; for (const int32_t* a : idx) { h_idx[type][r].assign(a, a + n); ++r; }
        for (; r < 4; ++r) h_idx[type][r].clear();
        roles_dirty = true;
    }
    void build_roles(int64_t cap) {
        static const int n_roles[4] = {2, 3, 4, 2};
        int64_t base = 0;
        for (int ty = 0; ty < 4; ++ty) { slot_base[ty] = base; base += (int64_t)n_roles[ty] * (int64_t)h_idx[ty][0].size(); }
        if (base >= ((int64_t)1 << 31)) throw ApiError{MHIP_ERR_CAPACITY, "too many specific-interaction slots"};
        n_slots = base;
        std::vector<int32_t> start(cap + 1, 0);
        for (int ty = 0; ty < 4; ++ty) for (int r = 0; r < n_roles[ty]; ++r) for (int32_t a : h_idx[ty][r]) ++start[a + 1];
        for (int64_t i = 0; i < cap; ++i) start[i + 1] += start[i];
        std::vector<int32_t> list((size_t)start[cap]);
        std::vector<int32_t> fill(start.begin(), start.end() - 1);
        for (int ty = 0; ty < 4; ++ty)
            for (int r = 0; r < n_roles[ty]; ++r)
                for (size_t t = 0; t < h_idx[ty][r].size(); ++t) list[(size_t)fill[h_idx[ty][r][t]]++] = (int32_t)(slot_base[ty] + (int64_t)n_roles[ty] * (int64_t)t + r);
        role_start.set(start.data(), start.size());
        if (list.empty()) list.push_back(0);
        role_slot.set(list.data(), list.size());
        if ((size_t)n_slots > slots_cap) { if (slots) (void)hipFree(slots); slots = nullptr; slots_cap = (size_t)n_slots; MHIP_HIP(hipMalloc((void**)&slots, slots_cap * sizeof(T4))); }
        roles_dirty = false; roles_cap = cap;
    }

    static void check(int64_t cap, int64_t n, std::initializer_list<const int32_t*> idx) {
        if (n < 0) throw ApiError{MHIP_ERR_INVALID, "negative interaction count"};
        for (const int32_t* a : idx) { if (n && !a) throw ApiError{MHIP_ERR_INVALID, "null index array"}; for (int64_t k = 0; k < n; ++k) if (a[k] < 0 || a[k] >= cap) throw ApiError{MHIP_ERR_INVALID, "specific interaction index out of range"}; }
    }
    void set_bonds(int64_t cap, int64_t n, const int32_t* i, const int32_t* j, const T* k, const T* r0) { check(cap, n, {i, j}); b_i.set(i, n); b_j.set(j, n); b_k.set(k, n); b_r0.set(r0, n); keep(0, n, {i, j}); }
    void set_angles(int64_t cap, int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const T* kth, const T* th0) {
        check(cap, n, {i, j, k}); a_i.set(i, n); a_j.set(j, n); a_k.set(k, n); a_kth.set(kth, n); a_th0.set(th0, n); keep(1, n, {i, j, k});
    }
    void set_torsions(int64_t cap, int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const int32_t* l, const int32_t* per, const T* ph, const T* k0) {
        check(cap, n, {i, j, k, l}); t_i.set(i, n); t_j.set(j, n); t_k.set(k, n); t_l.set(l, n); t_per.set(per, n); t_phase.set(ph, n); t_k0.set(k0, n); keep(2, n, {i, j, k, l});
    }
    void set_ewx(int64_t cap, int64_t n, const int32_t* i, const int32_t* j) { check(cap, n, {i, j}); x_i.set(i, n); x_j.set(j, n); keep(3, n, {i, j}); }
    void on_reorder() {}   // terms address atoms through inv[]: nothing to rebuild after a re-sort
    bool any() const { return b_i.n || a_i.n || t_i.n || x_i.n; }
    void release() { for (auto* h : {&b_i, &b_j, &a_i, &a_j, &a_k, &t_i, &t_j, &t_k, &t_l, &t_per, &x_i, &x_j}) h->release(); for (auto* h : {&b_k, &b_r0, &a_kth, &a_th0, &t_phase, &t_k0}) h->release(); role_start.release(); role_slot.release(); if (slots) (void)hipFree(slots); slots = nullptr; slots_cap = 0; }

    static constexpr int BT = 64;   // one wave per block: thousands of short, latency-bound terms spread over all CUs
    BondedArgs<T> args(const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv, T4* frc, double* part) const {
        BondedArgs<T> A;
        A.n_b = b_i.n; A.n_a = a_i.n; A.n_t = t_i.n; A.n_x = x_i.n;
        A.blk_b = cdiv(b_i.n, BT); A.blk_a = cdiv(a_i.n, BT); A.blk_t = cdiv(t_i.n, BT);
        A.b_i = b_i.p; A.b_j = b_j.p; A.b_k = b_k.p; A.b_r0 = b_r0.p;
        A.a_i = a_i.p; A.a_j = a_j.p; A.a_k = a_k.p; A.a_kth = a_kth.p; A.a_th0 = a_th0.p;
        A.t_i = t_i.p; A.t_j = t_j.p; A.t_k = t_k.p; A.t_l = t_l.p; A.t_per = t_per.p; A.t_phase = t_phase.p; A.t_k0 = t_k0.p;
        A.x_i = x_i.p; A.x_j = x_j.p; A.inv = inv; A.pos = pos; A.frc = frc; A.part = part; A.G = G; A.I = I;
        for (int ty = 0; ty < 4; ++ty) A.slot_base[ty] = slot_base[ty];
        return A;
    }
    int n_blocks() const { return cdiv(b_i.n, BT) + cdiv(a_i.n, BT) + cdiv(t_i.n, BT) + cdiv(x_i.n, BT); }
    // the slot path's tables for the current capacity (rebuilt after any set_*); args for the term kernel writing into the slots
    void ensure_roles(hipStream_t s, int64_t cap) { if (roles_dirty || roles_cap != cap) { MHIP_HIP(hipStreamSynchronize(s)); build_roles(cap); } }
    BondedArgs<T> slot_args(const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv) const { return args(G, I, pos, inv, slots, nullptr); }

    // forces ADDED to frc (sorted order; `orig` = sorted→caller map of the n_owned owned atoms, `cap` = context capacity).
    // (no atomics: a one-launch scatter with float atomics measured 2x slower and is not bit-reproducible)
    // terms_done: the term kernel's work was done by another launch already (forces_gs.hip runs the terms beside the pair groups): only the per-atom sums
    void launch_forces(hipStream_t s, const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv, T4* frc, const int32_t* orig, int64_t n_owned, int64_t cap, bool terms_done = false) {
        int nb = n_blocks();
        if (!nb) return;
        if (roles_dirty || roles_cap != cap) { MHIP_HIP(hipStreamSynchronize(s)); build_roles(cap); }
        if (!terms_done) hipLaunchKernelGGL((k_bonded<T, false, true>), dim3(nb), dim3(BT), 0, s, args(G, I, pos, inv, slots, nullptr));
        hipLaunchKernelGGL(k_bonded_collect<T>, dim3((unsigned)cdiv(n_owned * COLLECT_LANES, (int64_t)256)), dim3(256), 0, s, n_owned, orig, (const int32_t*)role_start.p, (const int32_t*)role_slot.p, (const T4*)slots, frc,
                           fold_parts, fold_n, fold_stride);
        fold_parts = nullptr; fold_n = 0;
        MHIP_HIP(hipGetLastError());
    }
    // partial force arrays the NEXT collect launch adds per atom (the group-split pair pass, forces_gs.hip); consumed by that launch
    const T4* fold_parts = nullptr; int fold_n = 0; int64_t fold_stride = 0;
    void fold(const T4* parts, int n, int64_t stride) { fold_parts = n > 0 ? parts : nullptr; fold_n = n; fold_stride = stride; }
    // nine component-major runs of per-block partial sums of the specific interactions' virial; returns the run length
    template <class DB> int launch_virial(hipStream_t s, const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv, DB& part) {
        int nb = n_blocks();
        if (!nb) return 0;
        part.reserve(9 * (size_t)nb);
        hipLaunchKernelGGL(k_bonded_virial<T>, dim3(nb), dim3(BT), 0, s, args(G, I, pos, inv, (T4*)nullptr, part.p), nb);
        MHIP_HIP(hipGetLastError());
        return nb;
    }
    // writes per-block partial energies into part (grown as needed); returns their count
    template <class DB> int launch_energy(hipStream_t s, const GridP<T>& G, const InterP<T>& I, const T4* pos, const int32_t* inv, DB& part) {
        int nb = n_blocks();
        if (!nb) return 0;
        part.reserve(nb);
        hipLaunchKernelGGL((k_bonded<T, true, false>), dim3(nb), dim3(BT), 0, s, args(G, I, pos, inv, (T4*)nullptr, part.p));
        MHIP_HIP(hipGetLastError());
        return nb;
    }
};

}  // namespace mhip

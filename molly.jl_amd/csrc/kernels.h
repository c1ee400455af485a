// kernels.h — the gfx950 device kernels of the engine (wave64, LDS-staged atom tiles).
//
// Design: atoms live in HBM sorted along a generalised Hilbert curve over a fine cell
// grid (cell ≈ r_list/2), as float4{x,y,z,q} + float2{σ,ϵ}.  A workgroup owns BI consecutive sorted atoms
// ("i-block").  At every neighbour rebuild k_build stages the block's spatial neighbourhood ("tile": every
// atom within r_list of the block's bounding box) in LDS, walks each i-atom's 5×5×5 cell stencil over the
// LDS tile with the reference's exact minimum-image arithmetic, and emits a per-atom FULL neighbour list of
// 16-bit tile slots (bit 15 = the reference's `special` flag), 4 entries per 8-byte row, interleaved so a
// wave reads 512 contiguous bytes per row.  Every force evaluation (k_forces) re-stages the tile from the
// current coordinates in block-local coordinates (periodic image resolved once per staged atom, not per
// pair), then each lane streams its rows and gathers partner atoms from LDS (ds_read_b128), accumulating its
// own force in registers: no atomics, no cross-lane reduction on the hot path, bit-reproducible forces.
#pragma once
#include <type_traits>
#include "physics.h"
#include "halo_xfer.h"
#include "philox.h"

namespace mhip {

enum { FLAG_MINIMG = 0, FLAG_OVERFLOW = 1, FLAG_MAX_TILE = 2, FLAG_MAX_ROWS = 3, FLAG_NAN = 4, FLAG_TOTAL_ROWS = 5, FLAG_MAX_CELLS = 6, FLAG_MAX_DISP2 = 7, FLAG_MAX_V2 = 8, N_FLAGS = 10 };
enum { OVF_TILE = 1, OVF_ROWS = 2, OVF_BOXCELLS = 4, OVF_SLOT = 8 };

constexpr int MAX_BOX_CELLS = 8192;
constexpr uint32_t XL_EXCLUDED = 1u << 30, XL_SPECIAL = 1u << 31, XL_INDEX = (1u << 30) - 1u;

// A list entry is 16 bits.  Two formats, fixed per list by the engine (`eshift`):
//   eshift = 0: bits 0-14 the tile slot, bit 15 the reference's `special` flag (every system with per-atom parameters or 1-4 pairs);
//   eshift = 2: the tile slot times four = the BYTE offset of the partner's x in the packed loop's x[] / y[] / z[] LDS arrays, no flag
//               bit (fp32 one-type LJ fluids, which have no special pairs; tiles of up to 16 383 atoms).  The hot loop then spends one
//               instruction per partner on its LDS address (v_and / v_lshrrev) instead of three (mask, shift, scale).
constexpr int ESHIFT_SCALED = 2;
constexpr int SLOT_MAX_SCALED = 16383;
__host__ __device__ inline uint32_t entry_slot(uint32_t e, int sh) { return sh ? (e >> sh) : (e & 0x7fffu); }
__host__ __device__ inline uint32_t entry_special(uint32_t e, int sh) { return sh ? 0u : (e >> 15); }
__host__ __device__ inline uint32_t make_entry(uint32_t slot, uint32_t sp, int sh) { return sh ? (slot << sh) : (slot | (sp << 15)); }

// ---------------------------------------------------------------------------------------------------
// small helpers
template <class T> __device__ inline int cell_coord(T x, int d, const GridP<T>& G) {
    // coordinates are stored wrapped into [0, L] on periodic axes (set_state / integrator wrap)
    T rel = G.periodic[d] ? x : x - G.origin[d];
    int c = (int)M<T>::floor(rel * G.inv_cs[d]);
    return min(max(c, 0), G.nc[d] - 1);
}

// lane `dst` (wave-uniform) of lo/hi := the two halves of a wave mask held in scalar registers (v_writelane_b32).  clang has no
// __builtin for it, so the LLVM intrinsic is bound by name: the compiler then knows the instruction and pads the gfx950 hazard between
// the v_cmp that produced the mask (a VALU write of VCC / an SGPR pair) and the VALU read of it as writelane data (2 wait states; a
// hand-written asm block is invisible to the hazard recogniser), and m0 stays the compiler's.
extern "C" __device__ int mhip_writelane_i32(int src, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ inline void mask_to_lane(int& lo, int& hi, unsigned long long mask, int dst) {
    lo = mhip_writelane_i32((int)(uint32_t)mask, dst, lo);
    hi = mhip_writelane_i32((int)(uint32_t)(mask >> 32), dst, hi);
}
// value of `v` in lane `src` (wave-uniform index) broadcast to every lane through the scalar file (v_readlane_b32)
__device__ inline float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ inline double lane_bcast(double v, int src) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <class T> __device__ inline T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}

// Sums and maxima over the 16-lane ROWS of a wave without the LDS: data-parallel-primitive moves (DPP) instead of ds_bpermute — a butterfly of
// __shfl_xor is six dependent LDS round trips per value, and in the epilogue of a pair pass the LDS pipe is what every other wave on the compute unit is
// busy with.  dpp_lane<C>: the value of another lane of the row (quad_perm xor 1 = 0xB1, xor 2 = 0x4E; row_ror:n = 0x120 + n: lane i reads lane i − n mod 16).
template <int CTRL> __device__ inline float dpp_lane(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ inline double dpp_lane(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Four sums at once: lanes 0..3 of every row return the row's sum of a, b, c, d respectively (the other lanes: partial sums of no use).  The association is
// the xor butterfly's — ((x0 + x1) + (x2 + x3)) in the quads, then (q0 + q1) + (q2 + q3) — so that (r0 + r1) + (r2 + r3) over the four rows gives, bit for
// bit, what `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)` leaves in every lane (the integrator kernels' partials, which these must agree with).
__device__ inline double row_sum4(double a, double b, double c, double d, int lane) {
    a += dpp_lane<0xB1>(a); b += dpp_lane<0xB1>(b); c += dpp_lane<0xB1>(c); d += dpp_lane<0xB1>(d);
    a += dpp_lane<0x4E>(a); b += dpp_lane<0x4E>(b); c += dpp_lane<0x4E>(c); d += dpp_lane<0x4E>(d);
    double s = (lane & 2) ? ((lane & 1) ? d : c) : ((lane & 1) ? b : a);
    s += dpp_lane<0x12C>(s);      // lane i + lane i + 4
    s += dpp_lane<0x128>(s);      // … + lanes i + 8, i + 12
    return s;
}
__device__ inline float row_max3(float a, float b, float c, int lane) {      // lanes 0..2 of every row: the row's maximum of a, b, c
    a = fmaxf(a, dpp_lane<0xB1>(a)); b = fmaxf(b, dpp_lane<0xB1>(b)); c = fmaxf(c, dpp_lane<0xB1>(c));
    a = fmaxf(a, dpp_lane<0x4E>(a)); b = fmaxf(b, dpp_lane<0x4E>(b)); c = fmaxf(c, dpp_lane<0x4E>(c));
    float s = (lane & 2) ? c : ((lane & 1) ? b : a);
    s = fmaxf(s, dpp_lane<0x12C>(s));
    s = fmaxf(s, dpp_lane<0x128>(s));
    return s;
}

// triclinic boxes: fractional coordinates of the upper-triangular basis (a ∥ x, b in the xy plane) and their height-scaled form u
template <class T> __device__ inline void frac_coords(T x, T y, T z, const GridP<T>& G, T s[3]) {
    s[2] = z * G.rs[2]; s[1] = (y - s[2] * G.bv[2][1]) * G.rs[1]; s[0] = (x - s[1] * G.bv[1][0] - s[2] * G.bv[2][0]) * G.rs[0];
}
// cell of a stored (wrapped) coordinate on the search grid, all three axes
template <class T> __device__ inline void cell_coords(T x, T y, T z, const GridP<T>& G, int c[3]) {
    if (G.tri_grid) {
        T s[3]; frac_coords(x, y, z, G, s);
#pragma unroll
        for (int d = 0; d < 3; ++d) { int q = (int)M<T>::floor(s[d] * G.hgt[d] * G.inv_cs[d]); c[d] = min(max(q, 0), G.nc[d] - 1); }
    } else { c[0] = cell_coord(x, 0, G); c[1] = cell_coord(y, 1, G); c[2] = cell_coord(z, 2, G); }
}
// Coordinates of a stored position relative to a block centre, nearest periodic image.  Cubic: ctr = the centre itself.  Triclinic
// grid: ctr = the centre's FRACTIONAL coordinates; the image is resolved in fractional space (rint per axis), the result is
// Cartesian.  For two atoms localised against the same centre the difference of the results is their true separation for that
// pair of images, and whenever it is shorter than half the smallest cell height it is the minimum image (any other image differs by
// a lattice vector, whose projection on some face normal is at least that height).
template <bool TRI, class T, class V4> __device__ inline void local_xyz_t(T& x, T& y, T& z, const V4& ctr, const GridP<T>& G);
template <class T, class V4> __device__ inline void local_xyz(T& x, T& y, T& z, const V4& ctr, const GridP<T>& G);

// x − c on a periodic axis, nearest image, for x and c wrapped into [0, L] (the image is then at k ∈ {−1, 0, +1} box lengths).  The two
// LARGE numbers are subtracted first — (L − c) when c is the one near L, (x − L) when x is: both differences are exact (Sterbenz) —
// so the result carries the rounding of a small number, not the ulp of L: 3.8e-6 nm in a 36 nm fp32 box, which is 7e-5 of the force
// of a contact pair across the periodic boundary (the reference's fp32 vector_1D, v = c2 − c1 first, carries exactly that error).
template <class T> __device__ inline T local_coord(T x, T c, T L, T invL) {
    T t = x - c;
    const T k = M<T>::rint(t * invL);
    if (k == T(-1)) t = x + (L - c);
    else if (k == T(1)) t = (x - L) - c;
    else if (k != T(0)) t -= L * k;
    return t;
}

// (TRI is a template argument where the call sits in a hot staging loop: with the branch inside the loop the pair kernel of the
// cubic 1M-atom fluid ran 12 % slower, 0.105 against 0.093 ms per pass)
template <bool TRI, class T, class V4> __device__ inline void local_xyz_t(T& x, T& y, T& z, const V4& ctr, const GridP<T>& G) {
    if constexpr (TRI) {
        // The fractional coordinates decide only WHICH image (three small integers); the local coordinates themselves are Cartesian differences against the
        // centre's Cartesian position — the same number for every atom localised against this centre, so it cancels in every pair — minus whole lattice vectors.
        // Round 6: going there and back through the fractional coordinates (s − ctr, × basis) cost an ulp of the CELL per atom — 1.5e-6 nm in a 13 nm fp32 cell,
        // 7.6e-5 of a contact pair's force (tools/micro/tri_xl_check.py) — where an atom that needs no shift now carries the rounding of a small number.
        T s[3]; frac_coords(x, y, z, G, s);
        const T n0 = M<T>::rint(s[0] - ctr.x), n1 = M<T>::rint(s[1] - ctr.y), n2 = M<T>::rint(s[2] - ctr.z);
        const T cx = ctr.x * G.bv[0][0] + ctr.y * G.bv[1][0] + ctr.z * G.bv[2][0], cy = ctr.y * G.bv[1][1] + ctr.z * G.bv[2][1], cz = ctr.z * G.bv[2][2];
        x = M<T>::fma(-n2, G.bv[2][0], M<T>::fma(-n1, G.bv[1][0], M<T>::fma(-n0, G.bv[0][0], x))) - cx;
        y = M<T>::fma(-n2, G.bv[2][1], M<T>::fma(-n1, G.bv[1][1], y)) - cy;
        z = M<T>::fma(-n2, G.bv[2][2], z) - cz;
    } else {
        x = G.periodic[0] ? local_coord(x, (T)ctr.x, G.L[0], G.invL[0]) : x - ctr.x;
        y = G.periodic[1] ? local_coord(y, (T)ctr.y, G.L[1], G.invL[1]) : y - ctr.y;
        z = G.periodic[2] ? local_coord(z, (T)ctr.z, G.L[2], G.invL[2]) : z - ctr.z;
    }
}
template <class T, class V4> __device__ inline void local_xyz(T& x, T& y, T& z, const V4& ctr, const GridP<T>& G) {
    if (G.tri_grid) local_xyz_t<true>(x, y, z, ctr, G); else local_xyz_t<false>(x, y, z, ctr, G);
}
// nearest-image displacement between two stored positions of the SAME atom (displacement checks: small vectors)
template <class T> __device__ inline void disp_image(T& ex, T& ey, T& ez, const GridP<T>& G) {
    if (G.triclinic) {
        T s[3]; frac_coords(ex, ey, ez, G, s);
        s[0] -= M<T>::rint(s[0]); s[1] -= M<T>::rint(s[1]); s[2] -= M<T>::rint(s[2]);
        ex = s[0] * G.bv[0][0] + s[1] * G.bv[1][0] + s[2] * G.bv[2][0]; ey = s[1] * G.bv[1][1] + s[2] * G.bv[2][1]; ez = s[2] * G.bv[2][2];
    } else {
        if (G.periodic[0]) ex -= G.L[0] * M<T>::rint(ex * G.invL[0]);
        if (G.periodic[1]) ey -= G.L[1] * M<T>::rint(ey * G.invL[1]);
        if (G.periodic[2]) ez -= G.L[2] * M<T>::rint(ez * G.invL[2]);
    }
}

// ---------------------------------------------------------------------------------------------------
// set_state / set_atoms: caller order → current sorted slots
// changed[0] / changed[1] (nullable) are raised when some coordinate / velocity differs from the one the engine holds: a state that
// is handed back unchanged (a run continued in chunks through get_state / set_state) then leaves lists and checks alone
template <class T>
__global__ void k_scatter_state(int64_t n_tot, int64_t n_owned, const int32_t* __restrict__ inv, const T* __restrict__ xyz,
                                const T* __restrict__ vxyz, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel, GridP<T> G, int32_t* changed) {
    int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool dx = false, dv = false;
    if (o < n_tot) {
        int s = inv[o];
        if (xyz) {
            T c[3] = {xyz[3 * o], xyz[3 * o + 1], xyz[3 * o + 2]};
            wrap_point(c[0], c[1], c[2], G);   // wrap_coords, simulators.jl:561
            const auto p = pos[s];
            dx = !(p.x == c[0] && p.y == c[1] && p.z == c[2]);
            pos[s].x = c[0]; pos[s].y = c[1]; pos[s].z = c[2];
        }
        if (vxyz && o < n_owned) {
            const auto v = vel[s];
            dv = !(v.x == vxyz[3 * o] && v.y == vxyz[3 * o + 1] && v.z == vxyz[3 * o + 2]);
            vel[s].x = vxyz[3 * o]; vel[s].y = vxyz[3 * o + 1]; vel[s].z = vxyz[3 * o + 2];
        }
    }
    if (changed) {      // (a word that is up already is left alone: 16 000 waves raising the same word one after the other took 0.3 of the launch's 0.36 ms at 1M atoms)
        const bool ax = __any(dx), av = __any(dv);
        if ((threadIdx.x & 63) == 0) {
            if (ax && __hip_atomic_load(&changed[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&changed[0], 1);
            if (av && __hip_atomic_load(&changed[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&changed[1], 1);
        }
    }
}

template <class T>
__global__ void k_scatter_params(int64_t n_tot, const int32_t* __restrict__ inv, const T* q, const T* sig, const T* eps,
                                 const T* mass, const T* lam, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel,
                                 typename Vec<T>::T2* lj) {
    int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (o >= n_tot) return;
    int s = inv[o];
    pos[s].w = q ? q[o] : T(0);
    T sg = sig ? sig[o] : T(0), ep = eps ? eps[o] : T(0);
    if (lam && lam[o] == T(0)) { sg = T(0); ep = T(0); }   // LJZeroShortcut on λ == 0 (mixing.jl:10)
    lj[s] = make2<T>(sg, ep);
    vel[s].w = mass ? mass[o] : T(1);
}

template <class T>
__global__ void k_gather_state(int64_t n, const int32_t* __restrict__ inv, const typename Vec<T>::T4* __restrict__ src, T* out) {
    int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (o >= n) return;
    auto p = src[inv[o]];
    out[3 * o] = p.x; out[3 * o + 1] = p.y; out[3 * o + 2] = p.z;
}

// forces: sorted → caller order (≙ reverse_reorder_forces_kernel!, kernels.jl:654-663)
template <class T>
__global__ void k_export_forces(int64_t n_owned, const int32_t* __restrict__ orig, const typename Vec<T>::T4* __restrict__ frc, T* out, int accumulate) {
    int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= n_owned) return;
    int o = orig[s];
    auto f = frc[s];
    if (accumulate) { out[3 * o] += f.x; out[3 * o + 1] += f.y; out[3 * o + 2] += f.z; }
    else { out[3 * o] = f.x; out[3 * o + 1] = f.y; out[3 * o + 2] = f.z; }
}

// ---------------------------------------------------------------------------------------------------
// rebuild step 1: sort key = Hilbert rank of the atom's cell (ghosts after all owned atoms) + histogram.
// Threads run over CALLER indices so that the stable radix sort orders the atoms of one cell by caller
// index: the sorted order — and with it every summation order — is a function of the coordinates alone,
// not of the history of earlier re-sorts.
template <class T>
__global__ void k_cell_keys(int64_t n_tot, int64_t n_owned, const typename Vec<T>::T4* __restrict__ pos, const int32_t* __restrict__ inv,
                            const uint32_t* __restrict__ cell_rank, uint32_t* key, int32_t* idx, int32_t* cell_cnt, GridP<T> G, int sub_bits) {
    int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (o >= n_tot) return;
    int s = inv[o];
    auto p = pos[s];
    int cc[3]; cell_coords(p.x, p.y, p.z, G, cc);
    uint32_t k = cell_rank[(cc[2] * G.nc[1] + cc[1]) * G.nc[0] + cc[0]];
    if (o >= n_owned) k += (uint32_t)G.ncell;
    // sub_bits (0 | 6): inside a cell the atoms are ordered along a 4×4×4 Morton curve instead of by caller index, so that neighbouring
    // lanes — and neighbouring tile slots — are neighbours in space (fewer distinct LDS banks per gather of the pair kernel)
    uint32_t sub = 0;
    if (sub_bits && !G.tri_grid) {
        const T xyz[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const T rel = (G.periodic[d] ? xyz[d] : xyz[d] - G.origin[d]) * G.inv_cs[d] - T(cc[d]);
            const int q = min(max((int)(rel * T(4)), 0), 3);
            sub |= (uint32_t)(q & 1) << d | (uint32_t)(q >> 1) << (3 + d);
        }
    }
    key[o] = (k << sub_bits) | sub; idx[o] = s;
    atomicAdd(&cell_cnt[k], 1);
}

// rebuild step 3: apply the sort permutation to every per-atom array (≙ reorder_system_kernel!, ext:1049-1067)
template <class T>
__global__ void k_permute(int64_t n_tot, const int32_t* __restrict__ perm, const typename Vec<T>::T4* __restrict__ pos_o,
                          const typename Vec<T>::T4* __restrict__ vel_o, const typename Vec<T>::T4* __restrict__ frc_o,
                          const typename Vec<T>::T2* __restrict__ lj_o, const int32_t* __restrict__ orig_o,
                          typename Vec<T>::T4* pos_n, typename Vec<T>::T4* vel_n, typename Vec<T>::T4* frc_n,
                          typename Vec<T>::T2* lj_n, int32_t* orig_n, int32_t* inv) {
    int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= n_tot) return;
    int so = perm[s];
    pos_n[s] = pos_o[so]; vel_n[s] = vel_o[so]; frc_n[s] = frc_o[so]; lj_n[s] = lj_o[so];
    int o = orig_o[so];
    orig_n[s] = o; inv[o] = (int32_t)s;
}

// ---------------------------------------------------------------------------------------------------
// rebuild step 4: tiles + neighbour lists, one workgroup (BI threads = BI/64 waves) per i-block
template <class T> struct BuildArgs {
    GridP<T> G;
    int64_t n_owned, n_tot;
    int BI, BI_shift, JS;            // i-atoms per block; waves sharing one i-atom (each takes every JS-th group of 64 tile atoms)
    int T_cap, R_cap, C_cap;         // capacities: tile atoms, list rows per (block, j-split), box cells
    const typename Vec<T>::T4* pos;
    const int32_t* orig;
    const int32_t* cell_start;       // 2*ncell+1: owned cells then ghost cells, in Hilbert-rank order
    const uint32_t* cell_rank;
    const int32_t* xl_start;         // exceptions, CSR over caller indices (null: none). Entry = partner caller index
    const uint32_t* xl_list;         //   | XL_EXCLUDED (bit 30) | XL_SPECIAL (bit 31); a pair in both lists is excluded
    int xl_span;                     // max |i - j| over all exception pairs: candidates further apart in caller index skip the scan
    int X_cap;                       // per-lane LDS capacity of the exception list
    int32_t* tile_idx;               // [n_blocks][T_cap] sorted slot of each tile atom
    int32_t* tile_cnt;               // [n_blocks]
    uint2* nbr;                      // [n_blocks][JS][R_cap][BI] 4×uint16 entries
    int32_t* wave_rows;              // [n_blocks][JS][BI/64]
    typename Vec<T>::T4* blk_center; // [n_blocks] centre (xyz) of the block's bounding box at build time
    int32_t* flags;
    T margin;
    int approx;                      // outer list of the dual scheme: any superset of r_list will do, skip the exact band test
    int walk;                        // search by walking every i-atom's cell stencil over the tile (1) or transposed, tile groups against the wave's i-atoms (0)
    int eshift;                      // entry format of the emitted rows (0 | ESHIFT_SCALED)
    uint16_t* cnt_out;               // [n_blocks][JS][BI] entries emitted per (j-split, atom), nullable (first lane order of the inner list)
    unsigned long long* dbg;         // builds with -DMHIP_STAMPS=1 (MOLLYHIP_DBG_TIMES): [block][wave][8] — wall clock at entry / behind the staging / behind the search / at the end, entries found, exception-list lengths
};

// exclusive prefix sum of a[0..n) in LDS, in place; a[n] receives the total.  `part` holds blockDim ints.
__device__ inline int block_excl_scan(int32_t* a, int n, int32_t* part, int tid, int nthr) {
    int per = (n + nthr - 1) / nthr, q0 = min(tid * per, n), q1 = min(q0 + per, n);
    int sum = 0;
    for (int q = q0; q < q1; ++q) sum += a[q];
    part[tid] = sum;
    __syncthreads();
    if (tid < WAVE) {   // first wave scans the per-thread sums (nthr <= 256)
        int run = 0;
        for (int base = 0; base < nthr; base += WAVE) {
            int v = (base + tid < nthr) ? part[base + tid] : 0, x = v;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) { int u = __shfl_up(x, o, WAVE); if (tid >= o) x += u; }
            if (base + tid < nthr) part[base + tid] = run + x - v;
            run += __shfl(x, WAVE - 1, WAVE);
        }
        if (tid == 0) a[n] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int q = q0; q < q1; ++q) { int v = a[q]; a[q] = run; run += v; }
    __syncthreads();
    return a[n];
}

// Thread-count limits of the block kernels (BI·JS lanes).  fp64 is held to 512 lanes so that a lane may use up to 256 VGPRs: at 1024
// lanes (128 VGPRs) every fp64 pair-kernel variant spilled to scratch.
template <class T> struct BlockLimits { static constexpr int max_threads = sizeof(T) == 8 ? 512 : 1024; };

// APPROX (the outer list of the dual pair list, a candidate set): everything inside r_list·(1 + 1e-4) is taken and the band is never
// re-decided with the reference's exact arithmetic — that path (a 27-image search, inlined per candidate) is then not even compiled:
// the kernel is half the size, and the walk variant fits its registers without spilling.
// XL: the system has exception lists (excluded / special pairs).  Without them the per-candidate lookup is not compiled either.
template <class T, bool WALK, bool APPROX, bool XL = true>
__global__ void __launch_bounds__(BlockLimits<T>::max_threads) k_build(BuildArgs<T> A) {
    using T4 = typename Vec<T>::T4;
    extern __shared__ __align__(32) unsigned char smem[];
    const GridP<T>& G = A.G;
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv_all = tid >> 6, NW_ALL = nthr >> 6;
    const int li = tid & (A.BI - 1), js = tid >> A.BI_shift;   // thread = (j-split, i-atom); every j-split group sees the same i-atoms
    const int wv = li >> 6, NW = A.BI >> 6;                    // i-wave of my atom, i-waves per block
#if MHIP_STAMPS
    if (A.dbg && lane == 0) { unsigned long long* d = A.dbg + ((size_t)b * NW_ALL + wv_all) * 8; d[0] = wall_clock64(); }
#endif
    // tile atoms in block-local coordinates, ALWAYS fp32: the search only needs them for the cheap pre-test, whose
    // 1e-4 band absorbs the rounding; decisions inside the band use the stored T coordinates from HBM
    // Three arrays, 12 bytes per tile atom (+ 4 for the caller index where exception lists need it): with 16-byte records + index the
    // 4 300-atom tile of the 1M-atom fluid's outer search took 104 KB and ONE 512-lane block fitted a CU — two waves per SIMD.
    float* t_x = reinterpret_cast<float*>(smem);
    float* t_y = t_x + A.T_cap;
    float* t_z = t_y + A.T_cap;
    int32_t* t_orig = reinterpret_cast<int32_t*>(t_z + A.T_cap);     // caller index of each tile atom (only with exception lists)
    int32_t* c_raw = t_orig + ((XL && A.xl_start) ? A.T_cap : 0);    // C_cap + 1: offsets of the cell-pruned candidate stream
    auto t_pos = [&](int t) { return make_float4(t_x[t], t_y[t], t_z[t], 0.f); };
    int32_t* c_rank = c_raw + (A.C_cap + 1);                  // C_cap: Hilbert rank of each box cell
    int32_t* part = c_rank + A.C_cap + 1;                     // nthr
    uint32_t* x_part = reinterpret_cast<uint32_t*>(part + nthr);   // [X_cap][BI] per-atom exception lists (only if xl_start)
    int32_t* t_off = reinterpret_cast<int32_t*>(x_part + (A.xl_start ? A.X_cap * A.BI : 0));   // C_cap + 1: first tile slot of each box cell (walk)
    __shared__ T s_sub[4][6];                                 // per-wave bounding boxes of the i-atoms
    __shared__ T s_ctr[3], s_half[3];
    __shared__ int s_boxlo[3], s_boxlen[3], s_full[3], s_exact, s_wtot[4 * 16];
    __shared__ int s_self[256];                                // tile slot of each i-atom itself (BI <= 256)

    // 0. bounding boxes: one per wave of i-atoms (tight pruning for elongated blocks) and their union
    const int64_t si = (int64_t)b * A.BI + li;
    const bool valid = si < A.n_owned;
    T4 pi = A.pos[valid ? si : (int64_t)b * A.BI];            // a block is never empty
    T my[3] = {pi.x, pi.y, pi.z};
    T mu[3] = {pi.x, pi.y, pi.z};                             // the same point in the frame of the cell grid (triclinic: u = s·h)
    if (G.tri_grid) { T sf[3]; frac_coords(my[0], my[1], my[2], G, sf); mu[0] = sf[0] * G.hgt[0]; mu[1] = sf[1] * G.hgt[1]; mu[2] = sf[2] * G.hgt[2]; }
    {
        T mn[3] = {mu[0], mu[1], mu[2]}, mx[3] = {mu[0], mu[1], mu[2]};
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int d = 0; d < 3; ++d) { T u = __shfl_xor(mn[d], o, WAVE); mn[d] = u < mn[d] ? u : mn[d]; T v = __shfl_xor(mx[d], o, WAVE); mx[d] = v > mx[d] ? v : mx[d]; }
        if (lane == 0 && js == 0) for (int d = 0; d < 3; ++d) { s_sub[wv][d] = mn[d]; s_sub[wv][3 + d] = mx[d]; }
    }
    __syncthreads();
    if (tid == 0) {
        int ovf = 0, exact = 0;
        for (int d = 0; d < 3; ++d) {
            T lo = s_sub[0][d], hi = s_sub[0][3 + d];
            for (int w = 1; w < NW; ++w) { lo = s_sub[w][d] < lo ? s_sub[w][d] : lo; hi = s_sub[w][3 + d] > hi ? s_sub[w][3 + d] : hi; }
            T half = (hi - lo) * T(0.5);
            s_ctr[d] = (lo + hi) * T(0.5); s_half[d] = half;
            s_full[d] = 0;
            if (G.all_cells[d]) { s_boxlo[d] = 0; s_boxlen[d] = G.nc[d]; s_full[d] = 1; if (G.periodic[d]) exact = 1; }
            else {
                T reach = G.r_list + A.margin;
                T rel_lo = (G.periodic[d] ? lo : lo - G.origin[d]) - reach, rel_hi = (G.periodic[d] ? hi : hi - G.origin[d]) + reach;
                int cl = (int)M<T>::floor(rel_lo * G.inv_cs[d]), ch = (int)M<T>::floor(rel_hi * G.inv_cs[d]);
                if (G.periodic[d]) {
                    int len = ch - cl + 1;
                    if (len >= G.nc[d]) { cl = 0; len = G.nc[d]; s_full[d] = 1; }
                    s_boxlo[d] = cl; s_boxlen[d] = len;
                    // block-local coordinates are unambiguous only if block half-extent + reach (+ drift) < L/2
                    if (s_full[d] || !(half + reach * T(1.25) < (G.tri_grid ? G.hgt[d] : G.L[d]) * T(0.5))) exact = 1;
                } else {
                    cl = max(cl, 0); ch = min(ch, G.nc[d] - 1);
                    s_boxlo[d] = cl; s_boxlen[d] = max(ch - cl + 1, 1);
                }
            }
        }
        if ((int64_t)s_boxlen[0] * s_boxlen[1] * s_boxlen[2] > A.C_cap) { ovf |= OVF_BOXCELLS; atomicMax(&A.flags[FLAG_MAX_CELLS], s_boxlen[0] * s_boxlen[1] * s_boxlen[2]); s_boxlen[0] = s_boxlen[1] = s_boxlen[2] = 1; }
        if (ovf) atomicOr(&A.flags[FLAG_OVERFLOW], ovf);
        if (exact) atomicOr(&A.flags[FLAG_MINIMG], 1);
        s_exact = exact;
        // what the pair kernel localises against: the centre itself, or (triclinic grid) its fractional coordinates
        if (G.tri_grid) A.blk_center[b] = make4<T>(s_ctr[0] / G.hgt[0], s_ctr[1] / G.hgt[1], s_ctr[2] / G.hgt[2], T(0));
        else A.blk_center[b] = make4<T>(s_ctr[0], s_ctr[1], s_ctr[2], T(0));
    }
    __syncthreads();
#if MHIP_STAMPS
    if (A.dbg && lane == 0) { unsigned long long* d = A.dbg + ((size_t)b * NW_ALL + wv_all) * 8; d[7] = wall_clock64(); }      // behind the boxes
#endif
    const int lx = s_boxlen[0], ly = s_boxlen[1], lz = s_boxlen[2];
    const int ncb = lx * ly * lz;
    const bool exact_only = s_exact != 0;
    const T ctr[3] = {s_ctr[0], s_ctr[1], s_ctr[2]};
    const T reach = G.no_list ? G.r_list : G.r_list + A.margin;
    const T reach2 = G.no_list ? G.r_list2 : reach * reach;

    // a stored position → loc: Cartesian coordinates relative to the block centre, nearest periodic image (the tile, the search);
    // ub: the same point in the frame of the wave boxes, relative to the centre (cubic: the same numbers; triclinic grid: u)
    const T cfrac[3] = {G.tri_grid ? ctr[0] / G.hgt[0] : T(0), G.tri_grid ? ctr[1] / G.hgt[1] : T(0), G.tri_grid ? ctr[2] / G.hgt[2] : T(0)};
    // (six scalars out, no arrays through pointers: with `T loc[3], T ub[3]` the two branches' stores were merged into stores through a pointer chosen at run
    // time, and the arrays lived in scratch — a store → load round trip per staged atom)
    struct Loc3 { T l0, l1, l2, u0, u1, u2; };
    auto localise3 = [&](T x, T y, T z) -> Loc3 {
        Loc3 r;
        if (G.tri_grid) {
            T sf[3]; frac_coords(x, y, z, G, sf);
            T f0 = sf[0] - cfrac[0]; f0 -= M<T>::rint(f0);
            T f1 = sf[1] - cfrac[1]; f1 -= M<T>::rint(f1);
            T f2 = sf[2] - cfrac[2]; f2 -= M<T>::rint(f2);
            r.u0 = f0 * G.hgt[0]; r.u1 = f1 * G.hgt[1]; r.u2 = f2 * G.hgt[2];
            r.l0 = f0 * G.bv[0][0] + f1 * G.bv[1][0] + f2 * G.bv[2][0]; r.l1 = f1 * G.bv[1][1] + f2 * G.bv[2][1]; r.l2 = f2 * G.bv[2][2];
        } else {
            T t0 = x - ctr[0]; if (G.periodic[0]) t0 -= G.L[0] * M<T>::rint(t0 * G.invL[0]);
            T t1 = y - ctr[1]; if (G.periodic[1]) t1 -= G.L[1] * M<T>::rint(t1 * G.invL[1]);
            T t2 = z - ctr[2]; if (G.periodic[2]) t2 -= G.L[2] * M<T>::rint(t2 * G.invL[2]);
            r.l0 = r.u0 = t0; r.l1 = r.u1 = t1; r.l2 = r.u2 = t2;
        }
        return r;
    };
    // squared distance from a point (frame of the boxes) to the nearest per-wave bounding box ("full" axes never prune).  On a triclinic
    // grid the axes of that frame are not orthogonal: each |Δu_d| is a lower bound of the distance on its own, so the test is the
    // LARGEST per-axis gap (Chebyshev), not their Euclidean sum.
    auto sub_dist2 = [&](T lx_, T ly_, T lz_) -> T {
        T best = T(3.0e38);
        T pc[3] = {lx_, ly_, lz_};
        for (int w = 0; w < NW; ++w) {
            T acc = T(0);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (s_full[d]) continue;
                T lo = s_sub[w][d] - ctr[d], hi = s_sub[w][3 + d] - ctr[d];
                T e = lo - pc[d]; T f = pc[d] - hi; e = e > f ? e : f; e = e > T(0) ? e : T(0);
                acc = G.tri_grid ? (e * e > acc ? e * e : acc) : acc + e * e;
            }
            best = acc < best ? acc : best;
        }
        return best;
    };

    // 1. cell-level pruning: per box cell the number of candidate atoms (0 if the cell is out of reach)
    const bool ghosts = A.n_tot > A.n_owned;
    for (int q = tid; q < ncb; q += nthr) {
        int qx = q % lx, qy = (q / lx) % ly, qz = q / (lx * ly);
        int g[3] = {s_boxlo[0] + qx, s_boxlo[1] + qy, s_boxlo[2] + qz};
        // cell AABB in the (unwrapped) frame of the bounding boxes, tested box-against-box
        T best = T(3.0e38);
        for (int w = 0; w < NW; ++w) {
            T acc = T(0);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (s_full[d]) continue;
                T clo = (G.periodic[d] ? T(0) : G.origin[d]) + T(g[d]) * G.cs[d], chi = clo + G.cs[d];
                T e = s_sub[w][d] - chi; T f = clo - s_sub[w][3 + d]; e = e > f ? e : f; e = e > T(0) ? e : T(0);
                acc = G.tri_grid ? (e * e > acc ? e * e : acc) : acc + e * e;
            }
            best = acc < best ? acc : best;
        }
        int gw[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { gw[d] = g[d]; if (G.periodic[d]) { gw[d] %= G.nc[d]; if (gw[d] < 0) gw[d] += G.nc[d]; } }
        int r = (int)A.cell_rank[(gw[2] * G.nc[1] + gw[1]) * G.nc[0] + gw[0]];
        bool keep = best <= reach2 * T(1.0001) + T(1e-12);
        int cnt = keep ? (A.cell_start[r + 1] - A.cell_start[r]) + (A.cell_start[G.ncell + r + 1] - A.cell_start[G.ncell + r]) : 0;
        c_raw[q] = cnt; c_rank[q] = ghosts ? r : A.cell_start[r];   // (no ghost atoms: the cell's first atom itself — the staging below then needs no lookup in global memory)
    }
    __syncthreads();
    const int nraw = block_excl_scan(c_raw, ncb, part, tid, nthr);
#if MHIP_STAMPS
    if (A.dbg && lane == 0 && !(XL && A.xl_start)) { unsigned long long* d = A.dbg + ((size_t)b * NW_ALL + wv_all) * 8; d[6] = wall_clock64(); }      // behind the cell pruning and its scan (systems without exception lists: slot 6 is theirs otherwise)
#endif

    // 2. atom-level pruning + ordered compaction into the LDS tile (cell-major, sorted order inside a cell)
    if constexpr (WALK) { for (int q = tid; q <= ncb; q += nthr) t_off[q] = 0; __syncthreads(); }
    int tile_n = 0;
    // Four raw atoms per lane and round, their (dependent) lookups issued together: cell by binary search in LDS, its first atom, the
    // atom's coordinates — three latencies in a row that, one atom per lane and two barriers per 512 atoms, made staging a quarter of
    // the whole search.  The compaction below keeps the raw order (sub-round by sub-round).
    constexpr int SUB = 4;
    for (int base = 0; base < nraw; base += SUB * nthr) {
        bool keep[SUB]; T4 p[SUB]; int s[SUB], q[SUB], og[SUB];
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const int t = base + u * nthr + tid;
            keep[u] = false; s[u] = 0; q[u] = 0; og[u] = 0;
            if (t < nraw) {
                int lo_q = 0, hi_q = ncb;             // largest q with c_raw[q] <= t
                while (hi_q - lo_q > 1) { int mid = (lo_q + hi_q) >> 1; if (c_raw[mid] <= t) lo_q = mid; else hi_q = mid; }
                q[u] = lo_q;
                int k = t - c_raw[lo_q], r = c_rank[lo_q];
                if (!ghosts) s[u] = r + k;
                else {
                    int own = A.cell_start[r + 1] - A.cell_start[r];
                    s[u] = k < own ? A.cell_start[r] + k : A.cell_start[G.ncell + r] + (k - own);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SUB; ++u) if (base + u * nthr + tid < nraw) { p[u] = A.pos[s[u]]; og[u] = (XL && A.xl_start) ? A.orig[s[u]] : 0; }
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            if (base + u * nthr + tid < nraw) {
                const Loc3 lc = localise3(p[u].x, p[u].y, p[u].z);
                // exact_only blocks (small boxes): images are ambiguous, keep the whole cell-pruned set
                keep[u] = exact_only ? true : sub_dist2(lc.u0, lc.u1, lc.u2) <= reach2;
                p[u].x = lc.l0; p[u].y = lc.l1; p[u].z = lc.l2;
            }
            const unsigned long long m = __ballot(keep[u]);
            if (lane == 0) s_wtot[u * NW_ALL + wv_all] = __popcll(m);
            q[u] = (int)((uint32_t)q[u] | ((uint32_t)__popcll(m & ((1ull << lane) - 1ull)) << 16));   // my rank inside the wave rides in the upper half
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            int dst = tile_n + (int)((uint32_t)q[u] >> 16);
            const int qc = q[u] & 0xffff;
            int tot = 0;
            for (int w = 0; w < NW_ALL; ++w) { const int c = s_wtot[u * NW_ALL + w]; if (w < wv_all) dst += c; tot += c; }
            if (keep[u] && dst < A.T_cap) {
                t_x[dst] = (float)p[u].x; t_y[dst] = (float)p[u].y; t_z[dst] = (float)p[u].z;
                if (XL && A.xl_start) t_orig[dst] = og[u];
                A.tile_idx[(int64_t)b * A.T_cap + dst] = s[u];
                const int64_t rel = (int64_t)s[u] - (int64_t)b * A.BI;
                if (rel >= 0 && rel < A.BI) s_self[rel] = dst;     // an i-atom always survives the pruning of its own block
                if constexpr (WALK) atomicAdd(&t_off[qc], 1);      // tile atoms per box cell (the compaction keeps the cell-major order)
            }
            tile_n += tot;
        }
        __syncthreads();
    }
    const int slot_max = A.eshift ? SLOT_MAX_SCALED : TILE_SLOT_MAX;
    if (tile_n > A.T_cap || tile_n > slot_max - 1) {
        if (tid == 0) { atomicOr(&A.flags[FLAG_OVERFLOW], tile_n > slot_max - 1 ? OVF_SLOT : OVF_TILE); atomicMax(&A.flags[FLAG_MAX_TILE], tile_n); A.tile_cnt[b] = 0; }
        if (lane == 0) A.wave_rows[(b * A.JS + js) * NW + wv] = 0;
        return;   // the host grows the capacities and rebuilds
    }
    if (tid == 0) A.tile_cnt[b] = tile_n;
    if (WALK && !exact_only) block_excl_scan(t_off, ncb, part, tid, nthr);   // → first tile slot of every box cell, t_off[ncb] = tile_n
#if MHIP_STAMPS
    if (A.dbg && lane == 0) { unsigned long long* d = A.dbg + ((size_t)b * NW_ALL + wv_all) * 8; d[1] = wall_clock64(); }
#endif

    // 3. neighbour search, transposed: the LANES hold 64 tile atoms (one coalesced LDS read per group), the wave
    //    loops over its own i-atoms on the scalar unit (coordinates broadcast with v_readlane), and one v_cmp per
    //    (i-atom, group) yields the 64 pair decisions as a wave mask in scalar registers, which a v_cndmask pair
    //    drops into lane i.  After 64 iterations lane i holds the bit mask of ITS neighbours in the group.  No
    //    divergence, no LDS latency, independent iterations.  Predicate of the reference:
    //      r2 = sum(abs2, vector(ci, cj, boundary)) <= r_list²  &&  eligible      (neighbors.jl:409-411)
    //    evaluated in block-local coordinates; only pairs within a 1e-4 band of r_list² — where rounding could
    //    change the outcome — are re-evaluated with the reference's exact arithmetic, so the emitted pair SET is
    //    bit-identical to the reference's.
    const uint32_t SENT = make_entry((uint32_t)tile_n, 0u, A.eshift);
    // four 16-bit entries per row and lane: `acc` holds the cnt & 3 entries of the row being filled, oldest in the low 16 bits; a row is stored when four are complete
    uint64_t acc = 0;
    int cnt = 0;
    uint2* my_rows = A.nbr + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    auto emit = [&](uint32_t e) {
        const int np = cnt & 3;
        acc |= (uint64_t)e << (np * 16);
        if (np == 3) { const int row = cnt >> 2; if (row < A.R_cap) my_rows[(int64_t)row * A.BI] = make_uint2((uint32_t)acc, (uint32_t)(acc >> 32)); acc = 0; }
        ++cnt;
    };
    // emit4 (the walk over an outer list's candidates, round 6): four candidates' entries at once, no branch per candidate — keep mask → byte-permute selectors
    // (16-entry LDS table) → the kept entries compacted, in order, to the low end of a 64-bit word → appended behind the cnt & 3 pending; at most one store per four
    // candidates.  The scheme of the pruning pass's emission (k_forces PRUNE); the entry-by-entry form cost a divergent branch, a 64-bit shift pair and a counter
    // test per CANDIDATE kept: 28 VALU instructions per candidate tested in the 1M-atom search (profiles/r06_traffic_lj1m_build.json)
    __shared__ uint2 s_lut[16];
    if (tid < 16) {
        uint32_t sel[2] = {0x0c0c0c0cu, 0x0c0c0c0cu};
        int p = 0;
        for (int j = 0; j < 4; ++j) if ((tid >> j) & 1) {
            const int sh = (p & 1) * 16;
            sel[p >> 1] = (sel[p >> 1] & ~(0xffffu << sh)) | ((uint32_t)((2 * j) | ((2 * j + 1) << 8)) << sh);
            ++p;
        }
        s_lut[tid] = make_uint2(sel[0], sel[1]);
    }
    [[maybe_unused]] auto emit4 = [&](uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3, bool k0, bool k1, bool k2, bool k3) {
        const uint32_t m = (k0 ? 1u : 0u) | (k1 ? 2u : 0u) | (k2 ? 4u : 0u) | (k3 ? 8u : 0u);
        const uint2 sel = s_lut[m];
        const uint32_t x = __builtin_amdgcn_perm(n1, n0, 0x05040100u), y = __builtin_amdgcn_perm(n3, n2, 0x05040100u);
        const uint64_t p = ((uint64_t)__builtin_amdgcn_perm(y, x, sel.y) << 32) | __builtin_amdgcn_perm(y, x, sel.x);
        const int np = cnt & 3, sh = np * 16;
        const uint64_t c_lo = acc | (p << sh), c_hi = (p >> 1) >> (63 - sh);     // (p >> (64 − sh), 0 for sh = 0)
        const int c = __builtin_popcount(m);
        const bool flush = np + c >= 4;
        if (flush) { const int row = cnt >> 2; if (row < A.R_cap) my_rows[(int64_t)row * A.BI] = make_uint2((uint32_t)c_lo, (uint32_t)(c_lo >> 32)); }
        acc = flush ? c_hi : c_lo;
        cnt += c;
    };
    __syncthreads();
    {
        const int oi = valid ? A.orig[si] : -1;
        int xl0 = 0, nxl = 0;
        if (XL && A.xl_start && valid) {   // my exception list: into LDS once (shared by the j-split group), scanned only for
            xl0 = A.xl_start[oi]; nxl = A.xl_start[oi + 1] - xl0;   // candidates close in caller index
            if (js == 0) for (int k = 0; k < min(nxl, A.X_cap); ++k) x_part[k * A.BI + li] = A.xl_list[xl0 + k];
        }
        if (XL && A.xl_start) __syncthreads();
        // My exceptions as two 128-bit masks over the caller-index offset oj − oi ∈ [−64, 63] (excluded / special): bonded partners sit next to each other in the
        // topology, so a candidate's verdict is one shift instead of a scan of the list — the time stamps of a 6mrr search (tools/build_times.py) showed the search
        // of a block growing with its atoms' list lengths (correlation 0.94: water 2 entries, 40 µs; protein interior 10, 93 µs), and the launch is one round of
        // blocks.  Partners further away (disulphide bridges, ring closures) set `far`, and only then the list itself is scanned.
        [[maybe_unused]] uint64_t mexc0 = 0, mexc1 = 0, mspc0 = 0, mspc1 = 0; [[maybe_unused]] bool far = false;
        if constexpr (XL) {
            for (int k = 0; k < nxl; ++k) {
                const uint32_t e = k < A.X_cap ? x_part[k * A.BI + li] : A.xl_list[xl0 + k];
                const int dl = (int)(e & XL_INDEX) - oi + 64;
                if ((unsigned)dl < 128u) {
                    const uint64_t bit = 1ull << (dl & 63);
                    if (e & XL_EXCLUDED) { if (dl < 64) mexc0 |= bit; else mexc1 |= bit; }
                    else if (e & XL_SPECIAL) { if (dl < 64) mspc0 |= bit; else mspc1 |= bit; }
                } else far = true;
            }
            mspc0 &= ~mexc0; mspc1 &= ~mexc1;      // (a pair in both lists is excluded)
        }
        // verdict for the candidate with caller index oj: −1 excluded, else its special flag
        [[maybe_unused]] auto exception_of = [&](int oj) -> int {
            const int dl = oj - oi + 64;
            if ((unsigned)dl < 128u) {
                const uint64_t me = dl < 64 ? mexc0 : mexc1, ms = dl < 64 ? mspc0 : mspc1; const int sh = dl & 63;
                return ((me >> sh) & 1ull) ? -1 : (int)((ms >> sh) & 1ull);
            }
            if (!far || (unsigned)(oj - oi + A.xl_span) > (unsigned)(2 * A.xl_span)) return 0;
            uint32_t hit = 0;
            for (int k = 0; k < nxl; ++k) {
                const uint32_t e = k < A.X_cap ? x_part[k * A.BI + li] : A.xl_list[xl0 + k];
                hit = ((e & XL_INDEX) == (uint32_t)oj) ? e : hit;
            }
            return (hit & XL_EXCLUDED) ? -1 : (int)(hit >> 31);
        };
#if MHIP_STAMPS
        if (A.dbg && XL && A.xl_start) { int x = nxl; for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE); if (lane == 0) A.dbg[((size_t)b * NW_ALL + wv_all) * 8 + 6] = (unsigned long long)x; }
#endif
        const Loc3 my_lc = localise3(my[0], my[1], my[2]);
        const float ml[3] = {(float)my_lc.l0, (float)my_lc.l1, (float)my_lc.l2};
        const float rl2 = G.no_list ? 3.0e38f : (float)G.r_list2;
        const float band_lo = rl2 * (1.0f - 1e-4f), band_hi = G.no_list ? 3.0e38f : rl2 * (1.0f + 1e-4f);
        const float reach2f = G.no_list ? 3.0e38f : (float)reach2;
        float blo[3], bhi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { blo[d] = (float)(s_sub[wv][d] - ctr[d]); bhi[d] = (float)(s_sub[wv][3 + d] - ctr[d]); }
        const unsigned long long valid_mask = __ballot(valid);
        const uint32_t self_t = valid ? (uint32_t)s_self[li] : 0xffffffffu;
        // this wave's share of the tile: the atoms t ≡ js (mod JS), a uniform sample in tile order, so that the JS
        // sub-lists of an i-atom come out equally long (little sentinel padding)
        // 3a. walk: every lane runs over the cells within `stencil` of its own atom's cell, row by row in x — the atoms of a row of cells are
        //     contiguous in the tile (cell-major, x fastest) — and tests each candidate itself.  (2·stencil + 1)³ cells hold about
        //     3.7 candidates per neighbour found; the transposed search below tests every i-atom of the wave against every 64-atom
        //     group of the tile that any of them can reach, about 10 per neighbour, and moves every hit through the scalar unit.
        //     Periodic axes never wrap inside a block's cell box here: a box that spans a whole periodic axis is an exact_only block.
        if (WALK && !exact_only) {
            const bool js_pow2 = (A.JS & (A.JS - 1)) == 0;
            if (valid) {
                int qc[3], lo[3], hi[3], mycell[3];
                cell_coords(my[0], my[1], my[2], G, mycell);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    int q = mycell[d] - s_boxlo[d];
                    if (G.periodic[d]) { if (q < 0) q += G.nc[d]; else if (q >= G.nc[d]) q -= G.nc[d]; }
                    qc[d] = q; lo[d] = max(q - G.stencil[d], 0); hi[d] = min(q + G.stencil[d], s_boxlen[d] - 1);
                }
                // (Measured in round 6 and not kept: trimming every row of cells to what the atom's OWN position inside its cell can reach — rows beyond reach skipped, the others cut
                // at their ends; it removes a third of a LANE's candidates and nothing of a WAVE's trips, whose count per row is the longest lane's: search phase of a block 60.2 →
                // 63.7 µs at 1M atoms, 6mrr's search 105 → 114 µs.)
                for (int qz = lo[2]; qz <= hi[2]; ++qz) for (int qy = lo[1]; qy <= hi[1]; ++qy) {
                    const int row = (qz * ly + qy) * lx;
                    const int t0 = t_off[row + lo[0]], t1 = t_off[row + hi[0] + 1];
                    // this j-split takes the slots ≡ js (mod JS); JS is a power of two for every shape the engine hands out (the general form
                    // costs two integer divisions per row of cells: 25 rows per atom)
                    int t = t0 + (js_pow2 ? ((js - t0) & (A.JS - 1)) : ((js - t0) % A.JS + A.JS) % A.JS);
                    auto consider = [&](int tc, const float4 pl) {
                        const float dx = pl.x - ml[0], dy = pl.y - ml[1], dz = pl.z - ml[2];
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        bool in = APPROX ? r2 <= band_hi : r2 < band_lo;
                        if (!APPROX && !in && r2 <= band_hi) {   // rare: decide with the reference's exact arithmetic on the stored coordinates
                            T4 pj = A.pos[A.tile_idx[(int64_t)b * A.T_cap + tc]];
                            T ex, ey, ez;
                            min_image_exact<T>(my[0], my[1], my[2], pj.x, pj.y, pj.z, G, ex, ey, ez);
                            in = norm2_exact(ex, ey, ez) <= G.r_list2;
                        }
                        if (!in || (uint32_t)tc == self_t) return;
                        uint32_t sp = 0;
                        const int oj = (XL && nxl > 0) ? t_orig[tc] : 0;
                        if constexpr (XL) {
                            if (nxl > 0) { const int v = exception_of(oj); if (v < 0) return; sp = (uint32_t)v; }
                        }
                        emit(make_entry((uint32_t)tc, sp, A.eshift));
                    };
                    // four candidates per round, their coordinates fetched together: with four waves per SIMD the LDS latency of one
                    // dependent read per candidate was what the walk waited for
                    const int last = t1 - 1;
                    if constexpr (APPROX && !XL) {
                        // a candidate set (the outer list) of a system without exception lists: no exact band, no lookups, so the four verdicts are plain compares and
                        // the kept entries go out together (emit4).  (With exception lists — 6mrr: 64-atom blocks, sixteen j-splits, two candidates per lane and row of
                        // cells — the four-wide form measured 7 % SLOWER than the entry-by-entry loop below, 113 against 105 µs per search: it stays on that loop.)
                        typedef float v2f __attribute__((ext_vector_type(2)));
                        const v2f mx2 = {ml[0], ml[0]}, my2 = {ml[1], ml[1]}, mz2 = {ml[2], ml[2]};
                        for (; t < t1; t += 4 * A.JS) {
                            const int ta = t, tb = t + A.JS, tc = t + 2 * A.JS, td = t + 3 * A.JS;
                            const int ib = min(tb, last), ic = min(tc, last), id = min(td, last);
                            const v2f dx0 = (v2f){t_x[ta], t_x[ib]} - mx2, dy0 = (v2f){t_y[ta], t_y[ib]} - my2, dz0 = (v2f){t_z[ta], t_z[ib]} - mz2;
                            const v2f dx1 = (v2f){t_x[ic], t_x[id]} - mx2, dy1 = (v2f){t_y[ic], t_y[id]} - my2, dz1 = (v2f){t_z[ic], t_z[id]} - mz2;
                            const v2f r20 = __builtin_elementwise_fma(dz0, dz0, __builtin_elementwise_fma(dy0, dy0, dx0 * dx0));
                            const v2f r21 = __builtin_elementwise_fma(dz1, dz1, __builtin_elementwise_fma(dy1, dy1, dx1 * dx1));
                            const bool ka = r20.x <= band_hi && (uint32_t)ta != self_t;
                            const bool kb = r20.y <= band_hi && (uint32_t)tb != self_t && tb < t1;
                            const bool kc = r21.x <= band_hi && (uint32_t)tc != self_t && tc < t1;
                            const bool kd = r21.y <= band_hi && (uint32_t)td != self_t && td < t1;
                            emit4(make_entry((uint32_t)ta, 0u, A.eshift), make_entry((uint32_t)ib, 0u, A.eshift), make_entry((uint32_t)ic, 0u, A.eshift), make_entry((uint32_t)id, 0u, A.eshift), ka, kb, kc, kd);
                        }
                    } else {
                    for (; t < t1; t += 4 * A.JS) {
                        const int ta = t, tb = t + A.JS, tc = t + 2 * A.JS, td = t + 3 * A.JS;
                        const float4 pa = t_pos(ta), pb = t_pos(min(tb, last)), pc = t_pos(min(tc, last)), pd = t_pos(min(td, last));
                        consider(ta, pa);
                        if (tb < t1) consider(tb, pb);
                        if (tc < t1) consider(tc, pc);
                        if (td < t1) consider(td, pd);
                    }
                    }
                }
                (void)qc;
            }
        } else {
        const int nwords = (tile_n + 64 * A.JS - 1) / (64 * A.JS);
        for (int w = 0; w < nwords; ++w) {
            const int jl = ((w << 6) + lane) * A.JS + js;
            float4 pl = make_float4(0.f, 0.f, 0.f, 0.f); bool near = false;
            T4 px = make4<T>(T(0), T(0), T(0), T(0));      // stored coordinates of my candidate (exact_only blocks)
            if (jl < tile_n) {
                pl = t_pos(jl);
                if (exact_only) { near = true; px = A.pos[A.tile_idx[(int64_t)b * A.T_cap + jl]]; }
                else {
                    float pc[3] = {pl.x, pl.y, pl.z}, acc = 0.f;
#pragma unroll
                    for (int d = 0; d < 3; ++d) { float e = blo[d] - pc[d]; float f = pc[d] - bhi[d]; e = e > f ? e : f; e = e > 0.f ? e : 0.f; acc += e * e; }
                    near = acc <= reach2f;
                }
            }
            if (__ballot(near) == 0ull) continue;          // the whole group is out of this wave's reach
            int mine_lo = 0, mine_hi = 0;                  // lane i: mask of its neighbours within this group
            if (!exact_only) {
                // which of my wave's i-atoms can reach this group at all?  bounding box of the group's near atoms
                // (wave min/max), every lane tests ITS i-atom against it; the scalar loop then visits only those.
                float gmn[3] = {near ? pl.x : 3.0e38f, near ? pl.y : 3.0e38f, near ? pl.z : 3.0e38f};
                float gmx[3] = {near ? pl.x : -3.0e38f, near ? pl.y : -3.0e38f, near ? pl.z : -3.0e38f};
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int d = 0; d < 3; ++d) { gmn[d] = fminf(gmn[d], __shfl_xor(gmn[d], o, WAVE)); gmx[d] = fmaxf(gmx[d], __shfl_xor(gmx[d], o, WAVE)); }
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) { float e = gmn[d] - ml[d]; float f = ml[d] - gmx[d]; e = e > f ? e : f; e = e > 0.f ? e : 0.f; acc += e * e; }
                unsigned long long im = __ballot(valid && acc <= reach2f);
                // candidates out of reach (or past the tile's end) are parked at infinity: no lane mask in the loop
                const float qx = near ? pl.x : __builtin_inff();
                if (APPROX) {
                    while (im) {
                        const int i = __builtin_ctzll(im);
                        im &= im - 1;
                        const float dx = qx - lane_bcast(ml[0], i), dy = pl.y - lane_bcast(ml[1], i), dz = pl.z - lane_bcast(ml[2], i);
                        mask_to_lane(mine_lo, mine_hi, __ballot(dx * dx + dy * dy + dz * dz <= band_hi), i);
                    }
                } else {
                    while (im) {
                        const int i = __builtin_ctzll(im);
                        im &= im - 1;
                        const float dx = qx - lane_bcast(ml[0], i), dy = pl.y - lane_bcast(ml[1], i), dz = pl.z - lane_bcast(ml[2], i);
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        unsigned long long in = __ballot(r2 < band_lo);
                        const unsigned long long maybe = __ballot(r2 <= band_hi) & ~in;
                        if (maybe) {   // rare: decide with the reference's exact arithmetic on the stored coordinates
                            const T ox = lane_bcast(my[0], i), oy = lane_bcast(my[1], i), oz = lane_bcast(my[2], i);
                            bool ok = false;
                            if ((maybe >> lane) & 1ull) {
                                T4 pj = A.pos[A.tile_idx[(int64_t)b * A.T_cap + jl]];
                                T ex, ey, ez;
                                min_image_exact<T>(ox, oy, oz, pj.x, pj.y, pj.z, G, ex, ey, ez);
                                ok = norm2_exact(ex, ey, ez) <= G.r_list2;
                            }
                            in |= __ballot(ok);
                        }
                        mask_to_lane(mine_lo, mine_hi, in, i);
                    }
                }
            } else {
                for (int i = 0; i < WAVE; ++i) {
                    const T ox = lane_bcast(my[0], i), oy = lane_bcast(my[1], i), oz = lane_bcast(my[2], i);
                    T ex, ey, ez;
                    min_image_exact<T>(ox, oy, oz, px.x, px.y, px.z, G, ex, ey, ez);
                    const unsigned long long in = __ballot(near && norm2_exact(ex, ey, ez) <= G.r_list2);
                    if (lane == i) { mine_lo = (int)(uint32_t)in; mine_hi = (int)(uint32_t)(in >> 32); }   // v_cndmask ×2
                }
            }
            // each lane unpacks its own mask (slot order = tile order: deterministic lists)
            unsigned long long mm = ((unsigned long long)(uint32_t)mine_hi << 32) | (uint32_t)mine_lo;
            if (!((valid_mask >> lane) & 1ull)) mm = 0;
            while (mm) {
                const int bit = __builtin_ctzll(mm);
                mm &= mm - 1;
                const uint32_t t = (uint32_t)(((w << 6) + bit) * A.JS + js);
                if (t == self_t) continue;                       // the atom itself (no LDS lookup on the common path)
                uint32_t sp = 0;
                const int oj = (XL && nxl > 0) ? t_orig[t] : 0;
                if constexpr (XL) {
                    if (nxl > 0) { const int v = exception_of(oj); if (v < 0) continue; sp = (uint32_t)v; }
                }
                emit(make_entry(t, sp, A.eshift));
            }
        }
        }   // transposed search
    }
#if MHIP_STAMPS
    if (A.dbg) {
        int c = cnt, x = 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, WAVE);
        if (lane == 0) { unsigned long long* d = A.dbg + ((size_t)b * NW_ALL + wv_all) * 8; d[2] = wall_clock64(); d[4] = (unsigned long long)c; d[5] = (unsigned long long)tile_n; (void)x; }
    }
#endif
    // 4. pad every lane to the wave's row count with the sentinel slot (a far-away dummy atom)
    if (A.cnt_out) A.cnt_out[((int64_t)b * A.JS + js) * A.BI + li] = (uint16_t)min(cnt, 65535);
    int rows_mine = (cnt + 3) >> 2;
    int rows_wave = wave_max(rows_mine);
    if (rows_wave > A.R_cap) { if (lane == 0) atomicOr(&A.flags[FLAG_OVERFLOW], OVF_ROWS); }
    const int rows_keep = rows_wave > A.R_cap ? 0 : rows_wave;
    while (((cnt + 3) >> 2) < rows_keep || (cnt & 3)) emit(SENT);
    if (lane == 0) A.wave_rows[(b * A.JS + js) * NW + wv] = rows_wave;   // > R_cap reports the required capacity; k_build_summary zeroes it
#if MHIP_STAMPS
    if (A.dbg && lane == 0) { unsigned long long* d = A.dbg + ((size_t)b * NW_ALL + wv_all) * 8; d[3] = wall_clock64(); }
#endif
}

// a[i] = b[i] = i
[[maybe_unused]] static __global__ void k_iota2(int64_t n, int32_t* a, int32_t* b) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { a[i] = (int32_t)i; b[i] = (int32_t)i; }
}

// flags[FLAG_NAN] != 0 unless every σ equals σ[0], every ϵ equals ϵ[0] and no λ is 0 (one-type system → uniform-LJ kernels)
template <class T>
__global__ void k_uniform_check(int64_t n, const T* __restrict__ sig, const T* __restrict__ eps, const T* __restrict__ lam, int32_t* flags) {
    const T s0 = sig[0], e0 = eps[0];
    bool bad = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad = bad || sig[i] != s0 || eps[i] != e0 || (lam && lam[i] == T(0));
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[FLAG_NAN], 1);
}

// reduce the per-block / per-wave results of k_build or k_filter into the flag words the host reads (flags zeroed before)
// out[c] = max of part[c·n .. (c + 1)·n), c = 0, 1, 2 — one block, plain stores (the three maxima of a validity check, k_vv_mid)
// (host: pinned host memory, nullable — the host reads it behind an event, no copy kernel in between)
// (swap01: out[0] and out[1] exchanged — the validity triple of a ghost plan wants {since the search, since the prune, v²}, the partials are {prune, search, v²})
[[maybe_unused]] static __global__ void k_track_reduce(int n, const float* __restrict__ part, float* out, float* host = nullptr, int swap01 = 0) {
    float m[3] = {0.f, 0.f, 0.f};
    constexpr int U = 4;      // (twelve loads in flight per lane: 4 096 partials by 256 lanes were sixteen memory latencies in a row)
    const int nt = (int)blockDim.x;
    for (int q = threadIdx.x; q < n; q += U * nt) {
        float v[3][U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const int i = min(q + k * nt, n - 1); v[0][k] = part[i]; v[1][k] = part[n + i]; v[2][k] = part[2 * n + i]; }
#pragma unroll
        for (int k = 0; k < U; ++k) { m[0] = fmaxf(m[0], v[0][k]); m[1] = fmaxf(m[1], v[1][k]); m[2] = fmaxf(m[2], v[2][k]); }
    }
    __shared__ float sh[3][16];
    for (int c = 0; c < 3; ++c) { m[c] = wave_max(m[c]); if ((threadIdx.x & 63) == 0) sh[c][threadIdx.x >> 6] = m[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float r = 0.f; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) r = fmaxf(r, sh[threadIdx.x][q]);
        const int o = (swap01 && threadIdx.x < 2) ? 1 - (int)threadIdx.x : (int)threadIdx.x;
        out[o] = r; if (host) host[o] = r;
    }
}

// The figures the host needs after a PRUNE pass — largest compacted tile, longest wave, rows in total, largest displacement since
// the outer search — by ONE block with plain stores, into the flag words AND into pinned host memory: no zeroing launch before it
// (k_build_summary accumulates with atomics), no copy launch behind it.  A wave whose rows overflowed R_cap is zeroed as there.
[[maybe_unused]] static __global__ void __launch_bounds__(1024) k_prune_summary(int n_blocks, int n_waves, int n_disp, int R_cap, const int32_t* __restrict__ tile_cnt, int32_t* wave_rows,
                                                                                const float* __restrict__ blk_disp2, int32_t* flags, int32_t* host_flags) {
    __shared__ int sh_t[16], sh_r[16], sh_s[16]; __shared__ float sh_d[16];
    int mt = 0, mr = 0, tot = 0; float md = 0.f;
    // (eight loads in flight per lane: one workgroup reads 50 000 words at a 1M-atom prune, and one load per trip was 52 memory latencies in a row — 25 µs)
    constexpr int U = 8;
    const int nt = (int)blockDim.x, t0 = (int)threadIdx.x;
    for (int q = t0; q < n_blocks; q += U * nt) {
        int v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = tile_cnt[min(q + k * nt, n_blocks - 1)];
#pragma unroll
        for (int k = 0; k < U; ++k) mt = max(mt, v[k]);
    }
    for (int q = t0; q < n_disp; q += U * nt) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = blk_disp2[min(q + k * nt, n_disp - 1)];
#pragma unroll
        for (int k = 0; k < U; ++k) md = fmaxf(md, v[k]);
    }
    for (int q = t0; q < n_waves; q += U * nt) {
        int v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = q + k * nt < n_waves ? wave_rows[q + k * nt] : 0;
#pragma unroll
        for (int k = 0; k < U; ++k) { const int r = v[k]; mr = max(mr, r); if (r > R_cap) wave_rows[q + k * nt] = 0; else tot += r; }      // (a slot past the end reads as 0 rows: neither branch does anything)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mt = max(mt, __shfl_xor(mt, o, 64)); mr = max(mr, __shfl_xor(mr, o, 64)); tot += __shfl_xor(tot, o, 64); md = fmaxf(md, __shfl_xor(md, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sh_t[threadIdx.x >> 6] = mt; sh_r[threadIdx.x >> 6] = mr; sh_s[threadIdx.x >> 6] = tot; sh_d[threadIdx.x >> 6] = md; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < (int)(blockDim.x >> 6); ++q) { mt = max(mt, sh_t[q]); mr = max(mr, sh_r[q]); tot += sh_s[q]; md = fmaxf(md, sh_d[q]); }
        int32_t out[N_FLAGS];
        for (int q = 0; q < N_FLAGS; ++q) out[q] = 0;
        out[FLAG_MAX_TILE] = mt; out[FLAG_MAX_ROWS] = mr; out[FLAG_TOTAL_ROWS] = tot; out[FLAG_MAX_DISP2] = (int32_t)__float_as_uint(md);
        for (int q = 0; q < N_FLAGS; ++q) { flags[q] = out[q]; host_flags[q] = out[q]; }
    }
}

[[maybe_unused]] static __global__ void k_build_summary(int n_blocks, int n_waves, int R_cap, const int32_t* __restrict__ tile_cnt, int32_t* wave_rows,
                                const float* __restrict__ blk_disp2, int32_t* flags) {
    __shared__ int sh_t[256], sh_r[256], sh_s[256]; __shared__ float sh_d[256];
    int mt = 0, mr = 0, tot = 0; float md = 0.f;
    const int gt = blockIdx.x * blockDim.x + threadIdx.x, gn = gridDim.x * blockDim.x;
    for (int q = gt; q < n_blocks; q += gn) { mt = max(mt, tile_cnt[q]); if (blk_disp2) md = fmaxf(md, blk_disp2[q]); }
    for (int q = gt; q < n_waves; q += gn) { int r = wave_rows[q]; mr = max(mr, r); if (r > R_cap) wave_rows[q] = 0; else tot += r; }
    sh_t[threadIdx.x] = mt; sh_r[threadIdx.x] = mr; sh_s[threadIdx.x] = tot; sh_d[threadIdx.x] = md;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh_t[threadIdx.x] = max(sh_t[threadIdx.x], sh_t[threadIdx.x + o]); sh_r[threadIdx.x] = max(sh_r[threadIdx.x], sh_r[threadIdx.x + o]);
            sh_s[threadIdx.x] += sh_s[threadIdx.x + o]; sh_d[threadIdx.x] = fmaxf(sh_d[threadIdx.x], sh_d[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(&flags[FLAG_MAX_TILE], sh_t[0]); atomicMax(&flags[FLAG_MAX_ROWS], sh_r[0]); atomicAdd(&flags[FLAG_TOTAL_ROWS], sh_s[0]);
        atomicMax(reinterpret_cast<unsigned int*>(&flags[FLAG_MAX_DISP2]), __float_as_uint(sh_d[0]));   // d2 >= 0: uint order = float order
    }
}

// ---------------------------------------------------------------------------------------------------
// max over n atoms of |x - x_snap|² (nearest image) → atomicMax into *out_word (float bits; d² >= 0 so uint order works); with
// `vel`, also the largest |v|² of the first n_vel atoms → *v2_word (how far anybody can get before the next check)
template <class T>
__global__ void k_max_disp(int64_t n, const typename Vec<T>::T4* __restrict__ pos, const typename Vec<T>::T4* __restrict__ snap, unsigned int* out_word, GridP<T> G,
                           const typename Vec<T>::T4* __restrict__ vel = nullptr, int64_t n_vel = 0, unsigned int* v2_word = nullptr,
                           const typename Vec<T>::T4* __restrict__ snap_b = nullptr, unsigned int* out_b = nullptr) {   // (a second snapshot in the same pass)
    float d2 = 0.f, v2 = 0.f, b2 = 0.f;
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto p = pos[s]; auto q = snap[s];
        T ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
        disp_image(ex, ey, ez, G);
        d2 = fmaxf(d2, (float)(ex * ex + ey * ey + ez * ez));
        if (snap_b) {
            q = snap_b[s];
            ex = p.x - q.x; ey = p.y - q.y; ez = p.z - q.z;
            disp_image(ex, ey, ez, G);
            b2 = fmaxf(b2, (float)(ex * ex + ey * ey + ez * ez));
        }
        if (vel && s < n_vel) { auto v = vel[s]; v2 = fmaxf(v2, (float)(v.x * v.x + v.y * v.y + v.z * v.z)); }
    }
    d2 = wave_max(d2); v2 = wave_max(v2); b2 = wave_max(b2);
    __shared__ float sh[16], shv[16], shb[16];   // (up to 1024 lanes per block: few blocks, few atomics on the result words)
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = d2; shv[threadIdx.x >> 6] = v2; shb[threadIdx.x >> 6] = b2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f, mv = 0.f, mb = 0.f;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) { m = fmaxf(m, sh[q]); mv = fmaxf(mv, shv[q]); mb = fmaxf(mb, shb[q]); }
        atomicMax(out_word, __float_as_uint(m));
        if (v2_word) atomicMax(v2_word, __float_as_uint(mv));
        if (out_b) atomicMax(out_b, __float_as_uint(mb));
    }
}

// ---------------------------------------------------------------------------------------------------
// Dual pair list ("dynamic pruning"): k_build searches with the OUTER radius r_list + Δ only every few rebuild
// intervals; at every rebuild step k_filter re-tests the outer entries at the current coordinates and keeps exactly
// the pairs with r2 <= r_list² (reference predicate, exact arithmetic inside the ±1e-4 band).  As long as no atom
// moved more than Δ/2 since the outer build — checked here, max displacement² goes to flags — the kept set is
// bit-identical to a fresh search.  Same tile, same slots, same row layout as the outer list.
template <class T> struct FilterArgs {
    GridP<T> G;
    int64_t n_owned;
    int BI, BI_shift, JS, T_cap, T_lds, R_cap, n_blocks;
    const typename Vec<T>::T4* pos;
    const typename Vec<T>::T4* pos_snap;      // coordinates at the outer build (same sorted order)
    const int32_t* tile_idx;
    const int32_t* tile_cnt;
    const uint2* nbr_out; const int32_t* rows_out;
    uint2* nbr_in; int32_t* rows_in;
    int32_t* tile_idx_in; int32_t* tile_cnt_in;   // compacted tile: only the atoms the inner list references
    const typename Vec<T>::T4* blk_center;
    float* blk_disp2;                         // [n_blocks] max displacement² of the block's atoms since the outer build
    int32_t* flags;
    T r_in, r_in2;                            // r_list and r_list² as the reference forms them (dist_cutoff ^ 2)
    int exact_all;                            // small boxes: block-local coordinates are ambiguous, decide every pair exactly
    int eshift;                               // entry format of both lists (0 | ESHIFT_SCALED)
};

template <class T>
__global__ void __launch_bounds__(BlockLimits<T>::max_threads) k_filter(FilterArgs<T> A) {
    using T4 = typename Vec<T>::T4;
    extern __shared__ __align__(32) unsigned char smem[];
    const GridP<T>& G = A.G;
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int li = tid & (A.BI - 1), js = tid >> A.BI_shift;
    const int tile_n = A.tile_cnt[b];
    float4* l_pos = reinterpret_cast<float4*>(smem);
    uint8_t* l_used = reinterpret_cast<uint8_t*>(l_pos + A.T_lds);            // byte t != 0: tile atom t is referenced by the inner list
    uint16_t* l_new = reinterpret_cast<uint16_t*>(l_used + ((A.T_cap + 8) & ~7));    // its slot in the compacted tile
    int32_t* l_part = reinterpret_cast<int32_t*>(l_new + ((A.T_cap + 2) & ~1));  // nthr scan scratch
    const T4 ctr = A.blk_center[b];
    const bool use_lds = !A.exact_all && tile_n <= A.T_lds;
    auto localise = [&](T4 p) -> float4 {
        local_xyz(p.x, p.y, p.z, ctr, G);
        return make_float4((float)p.x, (float)p.y, (float)p.z, 0.f);
    };
    const int32_t* tix = A.tile_idx + (int64_t)b * A.T_cap;
    if (use_lds) for (int t = tid; t < tile_n; t += nthr) l_pos[t] = localise(A.pos[tix[t]]);
    const int nuw = (tile_n + 3) >> 2;      // marks handled four at a time (32-bit words)
    for (int w = tid; w < nuw; w += nthr) reinterpret_cast<uint32_t*>(l_used)[w] = 0u;
    const int64_t si = (int64_t)b * A.BI + li;
    const bool valid = si < A.n_owned;
    const T4 pi = A.pos[valid ? si : (int64_t)b * A.BI];
    const float4 pil = localise(pi);
    // displacement of my atom since the outer build (nearest image): the host re-searches when 2·max > Δ
    __shared__ float s_d2[16];
    {
        float d2 = 0.f;
        if (valid && js == 0) {
            T4 q = A.pos_snap[si];
            T dx = pi.x - q.x, dy = pi.y - q.y, dz = pi.z - q.z;
            disp_image(dx, dy, dz, G);
            d2 = (float)(dx * dx + dy * dy + dz * dz);
        }
        d2 = wave_max(d2);
        if (lane == 0) s_d2[tid >> 6] = d2;
    }
    __syncthreads();
    if (tid == 0) { float m = 0.f; for (int w = 0; w < (nthr >> 6); ++w) m = fmaxf(m, s_d2[w]); A.blk_disp2[b] = m; }
    const int wslot = (b * A.JS + js) * (A.BI >> 6) + (li >> 6);
    const int rows = A.rows_out[wslot];
    const uint2* src = A.nbr_out + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    uint2* dst = A.nbr_in + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    const float rl2 = (float)A.r_in2, band_lo = rl2 * (1.0f - 1e-4f), band_hi = rl2 * (1.0f + 1e-4f);
    const uint32_t SENT = (uint32_t)tile_n;
    const int esh = A.eshift;
    uint32_t pack[2] = {0, 0};
    int cnt = 0;
    auto emit = [&](uint32_t e) {
        int k = cnt & 3;
        if (k == 0) { pack[0] = 0; pack[1] = 0; }
        pack[k >> 1] |= e << (16 * (k & 1));
        ++cnt;
        if (k == 3) dst[(int64_t)((cnt >> 2) - 1) * A.BI] = make_uint2(pack[0], pack[1]);
    };
    uint2 e_next = (0 < rows) ? src[0] : make_uint2(0, 0);
    for (int r = 0; r < rows; ++r) {
        const uint2 e4 = e_next;
        if (r + 1 < rows) e_next = src[(int64_t)(r + 1) * A.BI];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t e = ((k < 2 ? e4.x : e4.y) >> (16 * (k & 1))) & 0xffffu;
            uint32_t slot = entry_slot(e, esh);
            if (slot >= SENT || !valid) continue;
            bool in = false, maybe = true;
            if (use_lds) {
                float4 pj = l_pos[slot];
                float dx = pj.x - pil.x, dy = pj.y - pil.y, dz = pj.z - pil.z;
                float r2 = dx * dx + dy * dy + dz * dz;
                in = r2 < band_lo; maybe = !in && r2 <= band_hi;
            }
            if (maybe) {
                T4 pj = A.pos[tix[slot]];
                T ex, ey, ez;
                min_image_exact<T>(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, G, ex, ey, ez);
                in = norm2_exact(ex, ey, ez) <= A.r_in2;
            }
            if (in) { emit(e); l_used[slot] = 1; }   // benign race: every writer stores the same byte
        }
    }
    int rows_mine = (cnt + 3) >> 2;
    int rows_wave = wave_max(rows_mine);
    while (((cnt + 3) >> 2) < rows_wave || (cnt & 3)) emit(make_entry(SENT, 0u, esh));
    if (lane == 0) A.rows_in[wslot] = rows_wave;
    // compact the tile to the referenced atoms: rank of every used slot (ordered), new tile list, rows rewritten in place
    __syncthreads();
    {
        int per = (nuw + nthr - 1) / nthr, w0 = min(tid * per, nuw), w1 = min(w0 + per, nuw);
        const uint32_t* uw = reinterpret_cast<const uint32_t*>(l_used);
        int sum = 0;
        for (int w = w0; w < w1; ++w) sum += __popc(uw[w] & 0x01010101u);
        l_part[tid] = sum;
        __syncthreads();
        if (tid < WAVE) {
            int run = 0;
            for (int base = 0; base < nthr; base += WAVE) {
                int v = (base + tid < nthr) ? l_part[base + tid] : 0, x = v;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) { int u = __shfl_up(x, o, WAVE); if (tid >= o) x += u; }
                if (base + tid < nthr) l_part[base + tid] = run + x - v;
                run += __shfl(x, WAVE - 1, WAVE);
            }
            if (tid == 0) A.tile_cnt_in[b] = run;
        }
        __syncthreads();
        int run = l_part[tid];
        for (int w = w0; w < w1; ++w) {
            uint32_t m = uw[w];
#pragma unroll
            for (int q = 0; q < 4; ++q) if ((m >> (8 * q)) & 1u) { int t = (w << 2) + q; l_new[t] = (uint16_t)run; A.tile_idx_in[(int64_t)b * A.T_cap + run] = tix[t]; ++run; }
        }
    }
    __syncthreads();
    const uint32_t n_in = (uint32_t)A.tile_cnt_in[b];   // written by thread 0 before the barriers above
    for (int r = 0; r < rows_wave; ++r) {
        uint2 e4 = dst[(int64_t)r * A.BI];
        uint32_t out[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t e = ((k < 2 ? e4.x : e4.y) >> (16 * (k & 1))) & 0xffffu;
            uint32_t slot = entry_slot(e, esh);
            uint32_t ns = slot >= SENT ? n_in : (uint32_t)l_new[slot];
            out[k >> 1] |= make_entry(ns, entry_special(e, esh), esh) << (16 * (k & 1));
        }
        dst[(int64_t)r * A.BI] = make_uint2(out[0], out[1]);
    }
}

// ---------------------------------------------------------------------------------------------------
// the hot kernel: LJ + Coulomb forces (and energy) on the owned atoms from the LDS-staged tile
// The integrator's arithmetic is the reference's, operation by operation (no fused multiply-add, a true division): a = f / m with 0 for
// massless atoms (calc_accels, force.jl:17), v += a·dt/2 (simulators.jl:594, 616), x += v·dt (:602).  Every integrator kernel and the STEP
// epilogue of k_forces go through these two helpers, so a run cut into chunks repeats the uncut run bit for bit (test/simulation.jl:16-57).
template <class T> __device__ inline T accel_of(T f, T m) { return m == T(0) ? T(0) : f / m; }
template <class T> __device__ inline T step_add(T x, T rate, T h) { return M<T>::add(x, M<T>::mul(rate, h)); }

// ---- the fused step of a ghosted sub-domain (k_forces STEP + HALO, fp32 one-type fluids inside mhip_domain_run) --------------------------------------------
// ONE launch per MD step: the workgroups of blocks whose tile holds a ghost wait (bounded) for the peers' sequence words in their own prologue and stage the ghosts
// STRAIGHT from this rank's receive half (no unpack launch, no ghost slots in pos[]); every epilogue integrates its block's atoms (as on a single domain) and
// stores the new coordinates of the atoms the peers need — shifted — straight into the peers' receive halves (no pack launch); the last wave of the launch to
// finish sums the blocks' Σ m v partials into the message's centre-of-mass rows and raises this rank's sequence word at every peer.  The wire format is the one
// of k_halo_pack / k_halo_unpack (halo_xfer.h): a rank may take a fused step while a peer takes the separate launches.
// A word of a receive half, written by a PEER's stores (another device over xGMI, or another process on this one) and announced by its sequence word: read at system
// scope, so that what arrives does not depend on how the fine-grained region happens to be cached on this device (a plain load behind one wave's acquire is enough on a
// single device, where every writer goes through the same L2 — the only set-up this repository could run; round 6 made the reads explicit rather than find out on a node)
template <class U> __device__ inline U peer_row_load(const U* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
struct HaloSend { float* dst; float sx, sy, sz; int32_t pad; };      // one row an owned atom goes to: where in the peer's half (parity 0), with which periodic shift
struct HaloStep {
    const int32_t* order;            // [blocks_per_xcd · 8] block of workgroup w (−1: none): per XCD run the blocks WITHOUT ghosts first — the others wait for the peers
    const int32_t* flags;            // [n_blocks] bit 0: the block's tile holds a ghost; bit 1: some atom of the block is sent to a peer
    const int32_t* tsrc;             // [n_blocks · T_cap] the tile by source: >= 0 sorted slot of an owned atom (pos[]), < 0: row −1 − v of the receive half
    const float* rows;               // this exchange's half of my receive region (3 floats per row)
    const uint32_t* seq_in;          // my header's seq_in[parity of this exchange], indexed by rank
    const int32_t* peers; int n_peers; uint32_t seq_wait;
    int32_t* err; unsigned long long ticks;
    const int32_t* snd_start;        // [n_owned + 1] CSR over sorted slots → snd
    const HaloSend* snd;
    int64_t half_stride;             // floats between the two halves of a region
    int parity_send; uint32_t seq_send;
    uint32_t* const* ann;            // [n_peers] &peer's header.seq_in[0][my rank]
    unsigned int* done; unsigned int n_done;      // waves of i-atoms in the launch: the last to finish announces
    const int32_t* cm_row;           // [n_peers · cm_rows] rows of my half that carry peer p's {ΣPx, ΣPy, ΣPz, ΣM}
    float* const* cm_dst;            // [n_peers · cm_rows] where mine go in each peer's half (parity 0)
    int cm_rows; double* cm_all;     // [1 + n_peers][4]: [0] my own sums of the step before (k_halo_pack's cm_own, or the tail of the fused launch before)
};
// workgroup 0 of a fused ghosted launch: v_cm of the step before from my own sums and the peers' (their rows of this exchange) — the sum k_vv_mid's block_vcm
// forms over cm_all (entries in one wave, butterfly, P / M rounded to fp32) — published like step_cm_publish does
[[maybe_unused]] static __device__ inline void halo_cm_publish(const HaloStep& H, unsigned long long* pub, uint32_t seq, unsigned char* smem) {
    double (*tab)[4] = reinterpret_cast<double (*)[4]>(smem);
    const int tid = threadIdx.x;
    if (tid < H.n_peers && !xfer_wait(&H.seq_in[H.peers[tid]], H.seq_wait, H.err, H.ticks, (8 << 8) | H.peers[tid])) atomicOr(H.err, 1);
    __syncthreads();
    if (tid < 4) tab[0][tid] = H.cm_all[tid];
    if (tid < H.n_peers * 8) {      // four doubles = eight words, three per row
        const int p = tid >> 3, w = tid & 7;
        reinterpret_cast<float*>(tab[1 + p])[w] = peer_row_load(&H.rows[3 * (size_t)H.cm_row[p * H.cm_rows + w / 3] + w % 3]);
    }
    __syncthreads();
    if (tid < 64) {
        double a[4] = {0, 0, 0, 0};
        if (tid <= H.n_peers) for (int c = 0; c < 4; ++c) a[c] = tab[tid][c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], o, 64);
        if (tid < 3) {
            const double t = tid == 0 ? a[0] : tid == 1 ? a[1] : a[2];
            const float vc = (float)(t / a[3]);
            __hip_atomic_store(&pub[tid], (unsigned long long)__float_as_uint(vc) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// a block whose tile holds ghosts: wait for every sender's word before the staging reads their rows
[[maybe_unused]] static __device__ inline void halo_wait_block(const HaloStep& H) {
    if ((int)threadIdx.x < H.n_peers && !xfer_wait(&H.seq_in[H.peers[threadIdx.x]], H.seq_wait, H.err, H.ticks, (7 << 8) | H.peers[threadIdx.x])) atomicOr(H.err, 1);
    __syncthreads();
}
// The last wave of the launch to finish (every wave of i-atoms counts itself in behind its own stores): this rank's Σ m v of the step — the blocks' partials, read
// past the other XCDs' L2s — into cm_all[0] and into the centre-of-mass rows of every peer, then the sequence word: exchange seq_send is complete at the peers.
[[maybe_unused]] static __device__ inline void halo_tail(const HaloStep& H, const double* cm_part, int n_part) {
    const int lane = threadIdx.x & 63;
    double a[4] = {0, 0, 0, 0};
    if (cm_part) for (int q = lane; q < n_part; q += 64) for (int c = 0; c < 4; ++c) a[c] += __hip_atomic_load(&cm_part[4 * (int64_t)q + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], o, 64);
    if (lane < 4) H.cm_all[lane] = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : a[3];
    for (int q = lane; q < H.n_peers * H.cm_rows; q += 64) {
        const int r = q % H.cm_rows;
        float* d = H.cm_dst[q] + (int64_t)H.parity_send * H.half_stride;
        for (int c = 0; c < 3; ++c) {
            const int i = 3 * r + c;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(i < 2 ? a[0] : i < 4 ? a[1] : i < 6 ? a[2] : a[3]);
            const float w = i < 8 ? __uint_as_float((uint32_t)((i & 1) ? bits >> 32 : bits)) : 0.f;
            __hip_atomic_store(d + c, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_system();
    if (lane < H.n_peers) xfer_store_release(H.ann[lane] + (size_t)H.parity_send * XFER_MAX_RANKS, H.seq_send);
    if (lane == 0) __hip_atomic_store(H.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class T> struct ForceArgs {
    GridP<T> G;
    InterP<T> I;
    int64_t n_owned;
    int BI, BI_shift, JS, T_cap, T_lds, R_cap, n_blocks, blocks_per_xcd;   // T_lds: tile capacity of the LDS carve-up
    const typename Vec<T>::T4* pos;
    const typename Vec<T>::T2* lj;
    const int32_t* tile_idx;
    const int32_t* tile_cnt;
    const uint2* nbr;
    const int32_t* wave_rows;
    const typename Vec<T>::T4* blk_center;
    typename Vec<T>::T4* frc;
    double* pe_part;                 // [n_blocks] (ENERGY)
    // PRUNE pass of the dual pair list: the rows read above are the OUTER list; entries with r² <= r_prune2 are re-emitted as
    // the inner list, and the displacement of the block's atoms since the outer build is recorded for the host's validity check
    uint2* nbr_dst; int32_t* rows_dst; const typename Vec<T>::T4* pos_snap; float* blk_disp2; T r_prune2;
    // … and the tile is compacted to the atoms that can be referenced (slots renumbered as the rows are emitted)
    int32_t* tile_idx_dst; int32_t* tile_cnt_dst; int mark_off;   // byte offset of the renumbering table in dynamic LDS
    typename Vec<T>::T4* snap_dst;   // PRUNE: the coordinates this prune saw, for the next displacement checks (nullptr: the host copies them)
    int any_special;                 // 0: no special (1-4) pair exists, the per-entry weight select is compiled out (uniform-LJ fluids)
    int soa;                         // != 0: the packed fp32 one-type loop with the tile as x[] / y[] / z[] arrays `soa` dwords apart
    int eshift;                      // entry format of the rows read and written (0 | ESHIFT_SCALED)
    uint16_t* cnt_dst;               // PRUNE: [n_blocks][JS][BI] entries kept per (j-split, atom) — what k_regroup deals to the groups (nullable)
    // ghosted sub-domains: a pass over only the blocks whose tile holds no ghost atom (part 1: they can run while the ghost coordinates
    // are still on the wire) or only the others (part 2); 0 = every block
    const int32_t* blk_ghost; int part;
    int level_pairs;                 // PRUNE, two sub-lists per atom: level the two lanes' entry counts before padding
    // MOLLYHIP_DBG_TIMES (builds with -DMHIP_STAMPS=1 only): [n_blocks][waves][8] — shader clock and 100 MHz wall clock at kernel entry, behind the
    // staging barrier, behind the row walk and at the end, per wave
    unsigned long long* dbg;
    // Σ m v of the integrator launch before this pass, still per-block partials (k_vv_mid): workgroup 0 of this pass adds them up into ONE partial behind its own
    // work, so that the integrator launch behind it can run as many short blocks as it likes without each of them re-summing the partials first.  nullptr: nothing to do
    const double* cm_fin_in; int cm_fin_n; double* cm_fin_out;
    // STEP variants (the plain fp32 one-type passes inside mhip_vv_run): the velocity-Verlet update of the block's own atoms in the epilogue — second kick of this
    // step, first kick and drift of the next (what k_vv_mid does in a launch of its own) — into vel and into the OTHER position buffer (every block still reads
    // this step's positions for its tile); no force array is written.  cm_in / cm_n: Σ m v partials of the launch before, summed and published as v_cm by an extra
    // workgroup at the head of the grid (step_cm_publish), subtracted here one launch late as in k_vv_mid; cm_out: this launch's partials, one per block (nullable);
    // trk_part: per-block maxima for the validity check of the pair lists (nullable), against snap_a / snap_b.
    typename Vec<T>::T4* vel; typename Vec<T>::T4* pos_next; T dt, dt2;
    const double* cm_in; int cm_n; unsigned long long* cm_pub; uint32_t step_seq; double* cm_out;
    float* trk_part; const typename Vec<T>::T4* snap_a; const typename Vec<T>::T4* snap_b;
    HaloStep H;                      // HALO variants only
    // LANG variants (STEP with the Langevin-middle update of philox.h instead of the velocity-Verlet one: mhip_langevin_run on the fp32 one-type fluids): the noise
    // parameters of this step and the atoms' ORIGINAL indices (the Philox counter does not follow the Hilbert order)
    const int32_t* orig; StochP<T> S;
};
// STEP launches: the first workgroup of the grid does nothing but this — Σ m v of the launch before (n_part partials, the fixed order of cm_finalize_in_block)
// → v_cm = P / M rounded to T as block_vcm does → three words {value bits, launch number} that every other workgroup's epilogue polls (relaxed agent-scope
// atomics: no fence, no cache write-back; the epilogues come ≈ 15 µs after this has finished)
// (its scratch is the launch's dynamic LDS: a static __shared__ array would move the tile off LDS address 0, which the packed loop addresses from)
[[maybe_unused]] static __device__ inline void step_cm_publish(const double* __restrict__ part, int n_part, unsigned long long* pub, uint32_t seq, unsigned char* smem) {
    double (*sh_pub)[4] = reinterpret_cast<double (*)[4]>(smem);
    double a[4] = {0, 0, 0, 0};
    for (int q = threadIdx.x; q < n_part; q += blockDim.x) { const double* p = part + 4 * (int64_t)q; a[0] += p[0]; a[1] += p[1]; a[2] += p[2]; a[3] += p[3]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], o, 64);
    if ((threadIdx.x & 63) == 0) for (int c = 0; c < 4; ++c) sh_pub[threadIdx.x >> 6][c] = a[c];
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0, m = 0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) { t += sh_pub[q][threadIdx.x]; m += sh_pub[q][3]; }
        const float vc = (float)(t / m);
        __hip_atomic_store(&pub[threadIdx.x], (unsigned long long)__float_as_uint(vc) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// (runs behind everything else of workgroup 0: nothing of it is live in the pair loop — in front of it the packed loop spilled 12 bytes, as a call it needed a stack)
[[maybe_unused]] static __device__ inline void cm_finalize_in_block(const double* __restrict__ part, int n_part, double* out4, unsigned char* smem) {
    double a[4] = {0, 0, 0, 0};
    for (int q = threadIdx.x; q < n_part; q += blockDim.x) { const double* p = part + 4 * (int64_t)q; a[0] += p[0]; a[1] += p[1]; a[2] += p[2]; a[3] += p[3]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], o, 64);
    double* sh = reinterpret_cast<double*>(smem);
    if ((threadIdx.x & 63) == 0) for (int c = 0; c < 4; ++c) sh[4 * (threadIdx.x >> 6) + c] = a[c];
    __syncthreads();
    if (threadIdx.x < 4) { double t = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sh[4 * q + threadIdx.x]; out4[threadIdx.x] = t; }
    __syncthreads();
}
// strides the packed loop is compiled for (odd numbers of dwords): tiles of up to stride − 1 atoms, 12·stride bytes of LDS.  The
// smallest that holds the tile is used: 36 KiB leaves room for four 512-lane blocks per CU, 48 KiB for three (measured: −12 % per pass).
constexpr int SOA_STRIDES[3] = {2049, 3073, 4097};
// dynamic LDS a PRUNE pass needs behind its tile: the renumbering table (2 bytes per tile atom of a segment), scan scratch, eight boxes,
// the atoms' row counts (lane order) and the destination waves' row counts
// where the packed pruning pass keeps its renumbering table: right behind the three tile arrays of stride `soa` dwords (+ 64 bytes), as launch_pair_kernel carves it
__host__ __device__ constexpr int prune_mark_ct(int soa) { return (3 * soa * 4 + 64 + 15) & ~15; }
__host__ __device__ inline size_t prune_lds_bytes(int t_seg, int nthr) { return (size_t)(((t_seg + 8) & ~7) * 2) + ((size_t)nthr + 4) * 4 + 8 * 8 * 4 + ((size_t)nthr + 64) * 4 + 16 * 8 + 32; }   // (+ the 16 byte-permute selectors of the row compaction)

// (the plain fp32 one-type passes run four 512-lane blocks per CU = eight waves per SIMD, which takes <= 64 VGPRs: held by attribute)
#ifndef MHIP_FAST_MIN_WAVES
#define MHIP_FAST_MIN_WAVES 8
#endif
#ifndef MHIP_SB
#define MHIP_SB 4       // atoms per lane and staging round of the packed loop
#endif
template <class T, int LJM, int COULM, bool ENERGY, bool MINIMG, bool SEG, bool PRUNE>
constexpr int force_min_waves() { return (std::is_same<T, float>::value && LJM == LJ_DIST_UNIFORM && COULM == MHIP_COUL_NONE && !ENERGY && !MINIMG && !SEG && !PRUNE) ? MHIP_FAST_MIN_WAVES : 1; }

template <class T, int LJM, int COULM, bool ENERGY, bool MINIMG, bool SEG, bool PRUNE, int SOA_STRIDE = 4097, bool STEP = false, bool HALO = false, bool LANG = false>
__global__ void __launch_bounds__(BlockLimits<T>::max_threads, (force_min_waves<T, LJM, COULM, ENERGY, MINIMG, SEG, PRUNE>()))
k_forces(ForceArgs<T> A) {
    static_assert(!HALO || (STEP && std::is_same<T, float>::value), "the ghosted form exists for the fused fp32 step only");
    static_assert(!LANG || (STEP && !HALO), "the Langevin update rides in the fused single-domain step only");
    using T4 = typename Vec<T>::T4;
    using T2 = typename Vec<T>::T2;
    constexpr bool PER_ATOM_LJ = (LJM == LJ_DIST || LJM == LJ_GENERIC);
    extern __shared__ __align__(32) unsigned char smem[];
    const GridP<T>& G = A.G;
    // XCD-aware mapping: hardware places workgroup w on XCD w % 8; give each XCD a contiguous run of the
    // Hilbert-ordered blocks so that neighbouring blocks (which share most of their tiles) share an L2.
    int wg = blockIdx.x;
    if constexpr (STEP) {      // (the grid is one workgroup longer: the first one sums and publishes v_cm, the others take the blocks — XCD k + 1 gets the run XCD k had)
        if (wg == 0) {
            if constexpr (HALO) { if (A.cm_in) halo_cm_publish(A.H, A.cm_pub, A.step_seq, smem); }
            else if (A.cm_in) step_cm_publish(A.cm_in, A.cm_n, A.cm_pub, A.step_seq, smem);
            return;
        }
        --wg;
    }
    int b = (wg & 7) * A.blocks_per_xcd + (wg >> 3);
    [[maybe_unused]] int hflags = 0;
    if constexpr (HALO) { b = A.H.order[wg]; if (b < 0) return; hflags = A.H.flags[b]; }
    else {
        if (b >= A.n_blocks) return;
        if (A.part != 0 && (A.blk_ghost[b] != 0) != (A.part == 2)) return;
    }
    const int tid = threadIdx.x, nthr = blockDim.x;
    [[maybe_unused]] auto stamp = [&](int k) {
        if constexpr (MHIP_STAMPS != 0 && !PRUNE) {
            if (A.dbg && (tid & 63) == 0) { unsigned long long* d = A.dbg + ((size_t)b * (nthr >> 6) + (tid >> 6)) * 8; d[k] = __builtin_readcyclecounter(); d[4 + k] = wall_clock64(); }
        }
    };
    stamp(0);
    const int tile_n = A.tile_cnt[b];
    T4* l_pos = reinterpret_cast<T4*>(smem);
    T2* l_lj = reinterpret_cast<T2*>(l_pos + (A.T_lds + 1));
    const T4 ctr = A.blk_center[b];
    // only the fp32 one-type variants ever see the scaled entry format (the engine chooses it for them alone)
    constexpr bool MAY_SCALE = std::is_same<T, float>::value && LJM == LJ_DIST_UNIFORM && COULM == MHIP_COUL_NONE;
    const int esh = MAY_SCALE ? A.eshift : 0;

    auto localise = [&](T4 p, auto tri_tag) -> T4 {
        if constexpr (!MINIMG) local_xyz_t<decltype(tri_tag)::value>(p.x, p.y, p.z, ctr, G);
        return p;
    };
    // (the fp32 one-type kernels are kept free of the triclinic path — its mere presence cost them 12 % per pass at 1M atoms — and the
    // engine does not select them for a TriclinicBoundary)
    constexpr bool NO_TRI = std::is_same<T, float>::value && LJM == LJ_DIST_UNIFORM && COULM == MHIP_COUL_NONE && !ENERGY;
    const bool tri_local = !MINIMG && !NO_TRI && G.tri_grid;
    const int li = tid & (A.BI - 1), js = tid >> A.BI_shift;
    const int64_t si = (int64_t)b * A.BI + li;
    const bool valid = si < A.n_owned;
    // (STEP, whose lanes are in atom order: the atom again, from the lane number, where it is needed late — the compiler forms such addresses at the top of the
    // kernel otherwise, and carrying them past the pair loop costs the loop its 64 registers)
    [[maybe_unused]] auto step_lane = [&]() -> int { int t = tid; asm volatile("" : "+v"(t)); return t & (A.BI - 1); };
    [[maybe_unused]] auto step_atom = [&]() -> int64_t { return (int64_t)b * A.BI + step_lane(); };
    // What a lane needs of its own — its record, its row count — is ASKED FOR here and looked at behind the tile's staging (own_ready below).  Looked at here,
    // each was a memory latency of its own in front of the tile's index loads, which are one in front of the gathers: four dependent round trips of ≈ 2 µs under a
    // 3 TB/s list stream before a wave walked its first row, during which its block kept a quarter of a compute unit (profiles/r05_force_ab.txt §10).
    T4 pi_raw;
    if constexpr (PRUNE || COULM != MHIP_COUL_NONE) pi_raw = A.pos[valid ? si : (int64_t)b * A.BI];
    else {      // (no charge: three words — the fourth, unused, would be a register the compiler hands out again at once, and the write to it waits for the whole record)
        const T* own = reinterpret_cast<const T*>(A.pos + (valid ? si : (int64_t)b * A.BI));
        pi_raw = make4<T>(own[0], own[1], own[2], T(0));
    }
    T4 pi;
    T2 lji_raw = make2<T>(T(0), T(0)), lji = lji_raw;
    if constexpr (PER_ATOM_LJ) lji_raw = A.lj[valid ? si : (int64_t)b * A.BI];
    // fp32: the tile (and this lane's own record) carries √ϵ, 0 where σ = 0 — GeometricMixing + LJZeroShortcut become one product per pair
    constexpr bool PRE_E = PER_ATOM_LJ && sizeof(T) == 4;
    auto pre_e = [](T2 v) { if constexpr (PRE_E) { v.y = v.x == T(0) ? T(0) : M<T>::sqrt(v.y); v.x *= T(0.5); } return v; };   // (σ/2 as well: LorentzMixing becomes one add, and halving is exact)
    // this wave's own sub-list (the j-split was done by k_build); the row count is the same for all 64 lanes: a scalar
    const int rows_lane = A.wave_rows[(b * A.JS + js) * (A.BI >> 6) + (li >> 6)];
    int rows = 0;
    bool own_done = false;
    auto own_ready = [&]() {
        if (own_done) return;
        own_done = true;
        if constexpr (NO_TRI) pi = localise(pi_raw, std::false_type{});
        else pi = tri_local ? localise(pi_raw, std::true_type{}) : localise(pi_raw, std::false_type{});
        lji = pre_e(lji_raw);
        rows = __builtin_amdgcn_readfirstlane(rows_lane);
    };
    if constexpr (PRUNE) own_ready();      // (the prune's bounding boxes are made of the block-local coordinates, before the staging)
    const uint2* my_rows = A.nbr + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    const int32_t* tix = (HALO ? A.H.tsrc : A.tile_idx) + (int64_t)b * A.T_cap;
    if constexpr (HALO) { if (hflags & 1) halo_wait_block(A.H); }      // (behind the requests for the lane's own record and row count: they travel meanwhile)
    T fx = T(0), fy = T(0), fz = T(0), pe = T(0);
    [[maybe_unused]] T vir[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};   // ENERGY: Σ fr·(dx², dy², dz², dx·dy, dx·dz, dy·dz), the pair virial dr ⊗ f (force.jl:848-852)
    // PRUNE: inner-list emission state (same row format as k_build)
    uint64_t acc = 0;          // the kept & 3 entries of the row being filled, oldest in the low 16 bits
    int kept = 0;
    uint2* out_rows = nullptr;
    if constexpr (PRUNE) out_rows = A.nbr_dst + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    // PRUNE: renumbering table of the staged tile (segment): new slot of every tile atom that an inner-list entry of this block can
    // name, decided BEFORE the rows are walked — a tile atom further than the prune radius from the bounding box of every wave's
    // i-atoms is within the radius of none of them — so that the rows are emitted once, already renumbered (the marks taken during
    // the walk, as before, meant writing the rows, reading them back and writing them again: 1.9× the traffic of the pass)
    [[maybe_unused]] uint16_t* l_new = nullptr; [[maybe_unused]] int32_t* l_scan = nullptr; [[maybe_unused]] float* l_box = nullptr;
    [[maybe_unused]] int32_t* l_cnt = nullptr; [[maybe_unused]] int32_t* l_wmax = nullptr; [[maybe_unused]] const uint2* l_lut = nullptr;
    [[maybe_unused]] int n_new = 0;          // compacted tile atoms so far (block-uniform)
    if constexpr (PRUNE) {
        l_new = reinterpret_cast<uint16_t*>(smem + A.mark_off);
        l_scan = reinterpret_cast<int32_t*>(smem + A.mark_off + ((A.T_lds + 8) & ~7) * 2);
        l_box = reinterpret_cast<float*>(l_scan + nthr + 4);
        l_wmax = reinterpret_cast<int32_t*>(l_box + 64); l_cnt = l_wmax + 64;
        if (tid < 64) l_wmax[tid] = 0;
        // selectors of the row compaction: for each keep mask of a row's four entries, the v_perm_b32 byte selectors that move the kept
        // 16-bit entries, in order, to the low end of a 64-bit word (0x0c = a zero byte)
        l_lut = reinterpret_cast<const uint2*>(l_cnt + nthr);
        if (tid < 16) {
            uint32_t sel[2] = {0x0c0c0c0cu, 0x0c0c0c0cu};
            int p = 0;
            for (int j = 0; j < 4; ++j) if ((tid >> j) & 1) {
                const int sh = (p & 1) * 16;
                sel[p >> 1] = (sel[p >> 1] & ~(0xffffu << sh)) | ((uint32_t)((2 * j) | ((2 * j + 1) << 8)) << sh);
                ++p;
            }
            reinterpret_cast<uint2*>(l_cnt + nthr)[tid] = make_uint2(sel[0], sel[1]);
        }
        // bounding boxes of the i-atoms, eight per block (runs of BI/8 consecutive atoms: compact along the Hilbert curve), in the frame
        // of the staged tile (block-local coordinates).  One box per wave let half of a 64-atom water block's outer tile through.
        const int lpb = A.BI >> 3;
        float mn[3] = {(float)pi.x, (float)pi.y, (float)pi.z}, mx[3] = {(float)pi.x, (float)pi.y, (float)pi.z};
        for (int o = lpb >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, WAVE)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, WAVE)); }
        if (js == 0 && (li & (lpb - 1)) == 0) {    // a box = two 16-byte LDS words {lo.xyz, –}, {hi.xyz, –}
            reinterpret_cast<float4*>(l_box)[(li / lpb) * 2] = make_float4(mn[0], mn[1], mn[2], 0.f);
            reinterpret_cast<float4*>(l_box)[(li / lpb) * 2 + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
        }
        if (A.snap_dst && js == 0 && valid) A.snap_dst[si] = pi_raw;
    }
    // emit4: the four entries of an input row at once — keep mask → byte-permute selectors (LDS table) → the kept entries compacted to the
    // low end of a 64-bit word → appended behind the kept & 3 entries pending; a row is stored when four are complete (at most one
    // store per input row).  A 128-bit shift register with four v_alignbit + selects per ENTRY was 40 VALU instructions per row here.
    [[maybe_unused]] auto emit4 = [&](uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3, bool k0, bool k1, bool k2, bool k3) {
        const uint32_t m = (k0 ? 1u : 0u) | (k1 ? 2u : 0u) | (k2 ? 4u : 0u) | (k3 ? 8u : 0u);
        const uint2 sel = l_lut[m];
        const uint32_t x = __builtin_amdgcn_perm(n1, n0, 0x05040100u), y = __builtin_amdgcn_perm(n3, n2, 0x05040100u);   // the low halves only: a slot that is not kept may carry anything
        const uint64_t p = ((uint64_t)__builtin_amdgcn_perm(y, x, sel.y) << 32) | __builtin_amdgcn_perm(y, x, sel.x);
        const int np = kept & 3, sh = np * 16;
        const uint64_t c_lo = acc | (p << sh), c_hi = (p >> 1) >> (63 - sh);     // (p >> (64 − sh), 0 for sh = 0)
        const int cnt = __builtin_popcount(m);
        const bool flush = np + cnt >= 4;
        if (flush) out_rows[(int64_t)(kept >> 2) * A.BI] = make_uint2((uint32_t)c_lo, (uint32_t)(c_lo >> 32));
        acc = flush ? c_hi : c_lo;
        kept += cnt;
    };
    auto emit = [&](uint32_t e) {
        const int np = kept & 3;
        acc |= (uint64_t)e << (np * 16);
        if (np == 3) { out_rows[(int64_t)(kept >> 2) * A.BI] = make_uint2((uint32_t)acc, (uint32_t)(acc >> 32)); acc = 0; }
        ++kept;
    };

    constexpr bool FAST_CT = std::is_same<T, float>::value && LJM == LJ_DIST_UNIFORM && COULM == MHIP_COUL_NONE && !ENERGY && !MINIMG && !SEG;
    // The tile normally fits the LDS carve-up in one piece.  SEG: a tile larger than the LDS budget is
    // processed in segments; every segment re-walks the row stream and treats slots outside it as sentinels.
    const int seg_cap = A.T_lds;
    for (int seg_lo = 0; seg_lo < (SEG ? tile_n : 1); seg_lo += seg_cap) {
        const int n_here = SEG ? min(seg_cap, tile_n - seg_lo) : tile_n;
        if (SEG && seg_lo > 0) __syncthreads();
        // stage the tile: gathers of 16 B atoms (mostly L2 hits) into LDS, periodic image resolved once per staged atom.
        // The fp32 one-type loop needs x, y, z only and keeps them as three arrays SOA_STRIDE dwords apart (x[t], y[t], z[t]): each
        // component of two partners is then fetched by its own ds_read_b32 into the two halves of a 64-bit register and the whole pair
        // arithmetic is packed (v_pk_*_f32).  The odd stride keeps the LDS load/store optimiser from fusing the x and y reads of one
        // partner into a ds_read2(st64)_b32 — which would put (x, y) of ONE partner side by side and cost a transpose per component —
        // and spreads the three reads over different banks.
        const bool packed3 = FAST_CT && !A.any_special && A.soa == SOA_STRIDE && A.eshift == ESHIFT_SCALED;
        float* l_p3 = reinterpret_cast<float*>(smem);
        // SB atoms per lane and round, their dependent fetches issued together (slot → sorted index → coordinates: one atom per lane
        // and round left every round waiting for two memory latencies in a row, 10–20 µs per block at six rounds)
        auto stage = [&](auto tri_tag) {
            constexpr int SB = (FAST_CT && !PRUNE) ? MHIP_SB : 4;
            // The packed plain pass asks for the first round's tile indices WITHOUT knowing how many there are: tile_cnt[b] is a memory latency away, and a block keeps
            // its place on the compute unit while it waits (3.3 of the ≈ 20 µs a wave lives, profiles/r05_force_ab.txt §10).  The index loads are bounded by the
            // tile's capacity instead of its count, the first round runs unconditionally (a tile holds at least the block's own atoms), and the count is first needed
            // where the gathers are issued: lanes past the end fetch the block's first atom — ONE address, the gathers are what the staging is bound by — and write nothing.
            if constexpr (FAST_CT && !PRUNE) {
                auto round = [&](int t0) {
                    int s[SB]; T4 p[SB];
#pragma unroll
                    for (int k = 0; k < SB; ++k) s[k] = tix[min(t0 + k * nthr + tid, A.T_cap - 1)];
#pragma unroll
                    for (int k = 0; k < SB; ++k) {
                        s[k] = (t0 + k * nthr + tid < n_here) ? s[k] : (int)((int64_t)b * A.BI);
                        if constexpr (HALO) {      // a ghost comes from the receive half (three words of a 12-byte row), an owned atom from pos[]
                            if (s[k] >= 0) p[k] = A.pos[s[k]];
                            else { const float* gr = A.H.rows + 3 * (size_t)(-1 - s[k]); p[k] = make4<T>((T)peer_row_load(gr), (T)peer_row_load(gr + 1), (T)peer_row_load(gr + 2), T(0)); }
                        } else p[k] = A.pos[s[k]];
                    }
#pragma unroll
                    for (int k = 0; k < SB; ++k) {
                        const int t = t0 + k * nthr + tid;
                        if (t < n_here) {
                            const T4 pl = localise(p[k], tri_tag);
                            if (packed3) { l_p3[t] = (float)pl.x; l_p3[SOA_STRIDE + t] = (float)pl.y; l_p3[2 * SOA_STRIDE + t] = (float)pl.z; }
                            else l_pos[t] = pl;
                        }
                    }
                };
                round(0);      // (straight-line code: in front of a loop header the compiler drains every load in flight, the lane's own record among them)
                for (int t0 = SB * nthr; t0 < n_here; t0 += SB * nthr) round(t0);
                return;
            }
            for (int t0 = 0; t0 < n_here; t0 += SB * nthr) {
                int s[SB]; T4 p[SB]; [[maybe_unused]] T2 q[SB];
                // rounds of this batch that hold an atom at all (block-uniform; the packed plain pass goes without the test: with it the
                // register allocator spills 80 bytes per lane, and its tiles fill the rounds anyway)
                constexpr bool GUARD = !(FAST_CT && !PRUNE);
                const int kmax = GUARD ? min(SB, (n_here - t0 + nthr - 1) / nthr) : SB;
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    if (k < kmax) {
                        const int t = min(t0 + k * nthr + tid, n_here - 1);     // (clamped: the loads of a lane past the end are harmless duplicates)
                        s[k] = tix[seg_lo + t];
                    }
                }
#pragma unroll
                for (int k = 0; k < SB; ++k) if (k < kmax) { p[k] = A.pos[s[k]]; if constexpr (PER_ATOM_LJ) q[k] = A.lj[s[k]]; }
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const int t = t0 + k * nthr + tid;
                    if (k < kmax && t < n_here) {
                        const T4 pl = localise(p[k], tri_tag);
                        if (packed3) { l_p3[t] = (float)pl.x; l_p3[SOA_STRIDE + t] = (float)pl.y; l_p3[2 * SOA_STRIDE + t] = (float)pl.z; }
                        else l_pos[t] = pl;
                        if constexpr (PER_ATOM_LJ) l_lj[t] = pre_e(q[k]);
                    }
                }
            }
        };
        if constexpr (NO_TRI) stage(std::false_type{});
        else { if (tri_local) stage(std::true_type{}); else stage(std::false_type{}); }
        own_ready();
        if (tid == 0) {   // sentinel: far away (beyond every cutoff), no charge, no LJ
            if (packed3) { l_p3[n_here] = 1e4f; l_p3[SOA_STRIDE + n_here] = 1e4f; l_p3[2 * SOA_STRIDE + n_here] = 1e4f; }
            else l_pos[n_here] = make4<T>(T(1e4), T(1e4), T(1e4), T(0));
            if constexpr (PER_ATOM_LJ) l_lj[n_here] = make2<T>(T(0), T(0));
        }
        __syncthreads();
        if constexpr (PRUNE) {
            // which atoms of this segment stay in the compacted tile, and under which number (ordered: the compacted tile keeps the
            // cell-major order of the outer one)
            constexpr int NW = 8;                      // boxes per block
            const float reach2 = (float)A.r_prune2 * 1.001f + 1e-12f;
            // MINIMG passes (a block whose neighbourhood reaches half the box): tile and boxes are in stored, wrapped coordinates; the gap
            // to a box is taken per axis through the nearest image of the box centre (orthorhombic boxes: exact; the far-away sentinel
            // folds back into the box and is kept, which costs a slot).  Triclinic exact-image blocks keep everything, as before.
            const float bL[3] = {(float)G.L[0], (float)G.L[1], (float)G.L[2]}, biL[3] = {(float)G.invL[0], (float)G.invL[1], (float)G.invL[2]};
            auto stays = [&](int t) -> bool {
                if constexpr (MINIMG) { if (G.triclinic) return true; }
                float p[3];
                if (packed3) { p[0] = l_p3[t]; p[1] = l_p3[SOA_STRIDE + t]; p[2] = l_p3[2 * SOA_STRIDE + t]; }
                else { const T4 q = l_pos[t]; p[0] = (float)q.x; p[1] = (float)q.y; p[2] = (float)q.z; }
                float best = 3.0e38f;
#pragma unroll 1                                   // (unrolled, the sixteen box words are hoisted into 64 registers and the pruning variants spill)
                for (int w = 0; w < NW; ++w) {
                    const float4 lo = reinterpret_cast<const float4*>(l_box)[2 * w], hi = reinterpret_cast<const float4*>(l_box)[2 * w + 1];
                    float ex, ey, ez;
                    if constexpr (MINIMG) {
                        const float blo[3] = {lo.x, lo.y, lo.z}, bhi[3] = {hi.x, hi.y, hi.z};
                        float e[3];
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            float dd = p[d] - 0.5f * (blo[d] + bhi[d]);
                            if (G.periodic[d]) dd -= bL[d] * rintf(dd * biL[d]);
                            e[d] = fmaxf(fabsf(dd) - 0.5f * (bhi[d] - blo[d]) - 1e-5f * bL[d], 0.f);    // (a few ulp of the box length of slack: fp32 on stored coordinates)
                        }
                        ex = e[0]; ey = e[1]; ez = e[2];
                    } else {
                        ex = fmaxf(fmaxf(lo.x - p[0], p[0] - hi.x), 0.f); ey = fmaxf(fmaxf(lo.y - p[1], p[1] - hi.y), 0.f); ez = fmaxf(fmaxf(lo.z - p[2], p[2] - hi.z), 0.f);
                    }
                    best = fminf(best, ex * ex + ey * ey + ez * ez);
                }
                return best <= reach2;
            };
            const int per = (n_here + nthr - 1) / nthr, t0 = min(tid * per, n_here), t1 = min(t0 + per, n_here);
            int cnt = 0;
            uint32_t keep_bits = 0;                    // the verdicts of this lane's first 32 atoms (every lane has fewer in practice), for the second walk below
            for (int t = t0; t < t1; ++t) { const bool k = stays(t); cnt += k ? 1 : 0; if (t - t0 < 32) keep_bits |= (k ? 1u : 0u) << (t - t0); }
            l_scan[tid] = cnt;
            __syncthreads();
            if (tid < WAVE) {
                int run = 0;
                for (int base = 0; base < nthr; base += WAVE) {
                    int v = (base + tid < nthr) ? l_scan[base + tid] : 0, x = v;
#pragma unroll
                    for (int o = 1; o < WAVE; o <<= 1) { int u = __shfl_up(x, o, WAVE); if (tid >= o) x += u; }
                    if (base + tid < nthr) l_scan[base + tid] = run + x - v;
                    run += __shfl(x, WAVE - 1, WAVE);
                }
                if (tid == 0) l_scan[nthr] = run;
            }
            __syncthreads();
            {
                int run = n_new + l_scan[tid];
                for (int t = t0; t < t1; ++t) {
                    // (the packed loop reads the new slot as the ENTRY it emits — the slot's byte offset, slot << 2 — and saves the shift per entry)
                    if ((t - t0 < 32) ? ((keep_bits >> (t - t0)) & 1u) != 0u : stays(t)) { l_new[t] = (uint16_t)(packed3 ? run << ESHIFT_SCALED : run); A.tile_idx_dst[(int64_t)b * A.T_cap + run] = tix[seg_lo + t]; ++run; }
                    else l_new[t] = (uint16_t)0xffffu;
                }
            }
            n_new += l_scan[nthr];
            __syncthreads();
        }
        stamp(1);
        // the row stream is software-pipelined: row r+1 is in flight while row r is evaluated
        auto walk_rows = [&](auto spec_tag) {
            constexpr bool SPEC = decltype(spec_tag)::value;
            // One-type fp32 LJ fluid without special pairs (the 1M-atom benchmark): the pair arithmetic on float2 values, two partners
            // side by side, so that it maps onto v_pk_{add,mul,fma}_f32 without the register shuffles of the auto-vectorised generic
            // loop (forces_uniform.hip is compiled with the SLP vectoriser off).
            if constexpr (FAST_CT && !SPEC) { if (packed3) {
                // Two partners in the halves of 64-bit registers, component by component: (xa, xb), (ya, yb), (za, zb) come straight out of
                // three ds_read_b32 each, so the displacement, r², the LJ polynomial, the cutoff and the accumulation are ALL v_pk_*_f32
                // (two pairs per instruction).  Per two partners besides those: two address instructions (the entries ARE byte offsets:
                // v_and / v_lshrrev), two scalar products and ONE v_rcp_f32 (quarter rate) shared through 1/ra² = rb²·(1/(ra²rb²)).
                // F/r = (48ϵσ¹²/r⁶ − 24ϵσ⁶)/r⁸ with the two constants formed on the host.  The cutoff is a clamped
                // v_pk_fma instead of v_cmp + v_cndmask per partner: clamp((rc²⁺ − r²)·2¹⁰⁰) is exactly 1 for r² <= rc² and exactly 0
                // above (rc²⁺ = the float after rc²), the test of the reference (r <= rc).  A row's two packed chains are written side
                // by side so that each fills the other's wait states (a dependent read of a packed result costs one).
                typedef float v2f __attribute__((ext_vector_type(2)));
                const v2f pix = {(float)pi.x, (float)pi.x}, piy = {(float)pi.y, (float)pi.y}, piz = {(float)pi.z, (float)pi.z};   // (casts: the branch must also parse for T = double)
                const float c6 = (float)A.I.lj_c6, c12 = (float)A.I.lj_c12, rc2 = (float)A.I.lj_rc2, rp2 = (float)A.r_prune2;
                const float rc2n = __int_as_float(__float_as_int(rc2) + 1);
                const v2f cut_a = {-0x1p100f, -0x1p100f}, cut_b = {rc2n * 0x1p100f, rc2n * 0x1p100f};
                const v2f c48v = {c12, c12}, c24v = {c6, c6};
                v2f fx2 = {(float)fx, 0.f}, fy2 = {(float)fy, 0.f}, fz2 = {(float)fz, 0.f};
                // (the tile starts at LDS address 0 — these kernels have no static __shared__ — and saying so saves the base add per partner)
                typedef __attribute__((address_space(3))) const char* lds_cptr;
                const lds_cptr lbase = (lds_cptr)(uintptr_t)0;
                auto lds3 = [&](uint32_t oa, uint32_t ob, v2f& dx, v2f& dy, v2f& dz) {
                    typedef __attribute__((address_space(3))) const float* lds_fptr;
                    const lds_fptr pa = (lds_fptr)(lbase + oa), pb = (lds_fptr)(lbase + ob);
                    dx = (v2f){pa[0], pb[0]} - pix; dy = (v2f){pa[SOA_STRIDE], pb[SOA_STRIDE]} - piy; dz = (v2f){pa[2 * SOA_STRIDE], pb[2 * SOA_STRIDE]} - piz;
                };
                auto row = [&](const uint2 e4) {
                    const uint32_t oa = e4.x & 0xffffu, ob = e4.x >> 16, oc = e4.y & 0xffffu, od = e4.y >> 16;
                    v2f dx0, dy0, dz0, dx1, dy1, dz1;
                    lds3(oa, ob, dx0, dy0, dz0); lds3(oc, od, dx1, dy1, dz1);
                    // (the row's twelve ds_read_b32 in one run: left to itself the scheduler does that too — until a change somewhere else in the kernel shifts its
                    // register estimate, and then it waits for the reads pair by pair: +40-65 % per walk, profiles/r05_force_ab.txt §9.  Pinned, the trip's wait pattern
                    // is the same in every variant)
                    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
                    const v2f r20 = __builtin_elementwise_fma(dz0, dz0, __builtin_elementwise_fma(dy0, dy0, dx0 * dx0));
                    const v2f r21 = __builtin_elementwise_fma(dz1, dz1, __builtin_elementwise_fma(dy1, dy1, dx1 * dx1));
                    if constexpr (PRUNE) {   // (no slot test: the sentinel slot — padding, and everything a lane without an atom holds — lies 10⁴ nm away)
                        // the four new slot numbers are fetched up front, with the coordinates: looked up inside the branches, each kept
                        // entry waited for an LDS round trip of its own
                        // (round 6: the table sits at a compile-time LDS address behind the three tile arrays — A.mark_off, which the engine forms from the same numbers and
                        // launch_pair_kernel checks — so an entry's look-up is ONE shift and a read with an immediate offset, and the table holds the new entries themselves:
                        // 12 VALU instructions fewer per row of four than base + ((entry >> 2) << 1), read, << 2)
                        typedef __attribute__((address_space(3))) const unsigned short* lds_hptr;
                        auto new_entry = [&](uint32_t o) -> uint32_t { return (uint32_t)*(lds_hptr)(lbase + prune_mark_ct(SOA_STRIDE) + (o >> 1)); };
                        const uint32_t na = new_entry(oa), nb = new_entry(ob), nc = new_entry(oc), nd = new_entry(od);
                        emit4(na, nb, nc, nd, r20.x <= rp2, r20.y <= rp2, r21.x <= rp2, r21.y <= rp2);
                    }
                    const float t0 = __builtin_amdgcn_rcpf(r20.x * r20.y), t1 = __builtin_amdgcn_rcpf(r21.x * r21.y);
                    const v2f u0 = (v2f){r20.y, r20.x} * t0, u1 = (v2f){r21.y, r21.x} * t1;       // 1/r² of each partner
                    v2f in0, in1;
                    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp\n\ts_nop 0" : "=v"(in0) : "v"(r20), "s"(cut_a), "v"(cut_b));   // (the nop: a dependent read of a packed result needs one wait state, and asm is invisible to the hazard recogniser)
                    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp\n\ts_nop 0" : "=v"(in1) : "v"(r21), "s"(cut_a), "v"(cut_b));
                    const v2f q0 = u0 * u0, q1 = u1 * u1;
                    const v2f w0 = u0 * in0, w1 = u1 * in1;
                    const v2f c0 = q0 * u0, c1 = q1 * u1;                                        // 1/r⁶
                    const v2f g0 = __builtin_elementwise_fma(c0, c48v, -c24v), g1 = __builtin_elementwise_fma(c1, c48v, -c24v);
                    const v2f h0 = c0 * w0, h1 = c1 * w1;
                    const v2f f0 = g0 * h0, f1 = g1 * h1;                                        // (48ϵσ¹²/r⁶ − 24ϵσ⁶)/r⁸ inside the cutoff, else 0
                    fx2 -= dx0 * f0; fy2 -= dy0 * f0; fz2 -= dz0 * f0;
                    fx2 -= dx1 * f1; fy2 -= dy1 * f1; fz2 -= dz1 * f1;
                };
                // Two rows per trip, each fetched two rows ahead of its use, loop control on the scalar unit (three and four rows in flight, and
                // first rows requested before the staging, spill under the 64-VGPR bound: +4 … +10 % per pass, profiles/r03_force_ab.txt).  The
                // fetches are unconditional — the index is clamped to the last row, a few redundant loads per lane — so that the compiler can
                // count them: behind a branch it waits for ALL outstanding loads, the one it has just issued included.
                if (rows > 0) {
                    const int last = rows - 1;
                    constexpr int D = 2;
                    uint2 e[D] = {my_rows[0], my_rows[(int64_t)min(1, last) * A.BI]};
                    auto next = [&](int q) -> uint2 { return my_rows[(int64_t)min(q, last) * A.BI]; };
                    int r = 0;
                    for (; r + D <= rows; r += D) {
#pragma unroll
                        for (int k = 0; k < D; ++k) { row(e[k]); e[k] = next(r + D + k); }
                    }
#pragma unroll
                    for (int k = 0; k < D - 1; ++k) if (r + k < rows) row(e[k]);
                }
                fx = (T)(fx2.x + fx2.y); fy = (T)(fy2.x + fy2.y); fz = (T)(fz2.x + fz2.y);
                return;
            } }
            // fp32, per-atom σ / ϵ with the distance cutoff, reaction field or Ewald direct space (A&S erfc) — the solvated-protein loops:
            // two entries of a row side by side (pair_eval2, physics.h).  The gathers stay 16- and 8-byte records (a random
            // ds_read_b128 + ds_read_b64 costs 2.7 units of LDS time per partner, six ds_read_b32 of a structure-of-arrays tile 6): the
            // displacement is taken per partner — (dx, dy) as one packed subtraction out of the record's first register pair —, r² and
            // the per-partner parameters land in the halves of register pairs, and everything behind them is packed.  A row that names a
            // special (1-4) pair in any lane of the wave takes the one-partner loop below (3 094 of 4.6 M pairs in 6mrr).
            constexpr bool PK2 = std::is_same<T, float>::value && LJM == LJ_DIST && (COULM == MHIP_COUL_EWALD_DIRECT || COULM == MHIP_COUL_REACTION_FIELD) && !ENERGY;
            [[maybe_unused]] bool pk2_ok = false;
            if constexpr (PK2) pk2_ok = Pk2Consts::usable(reinterpret_cast<const InterP<float>&>(A.I));
            [[maybe_unused]] auto row_pk2 = [&](const uint2 e4) {
                if constexpr (PK2) {
                    const Pk2Consts K(A.I);
                    const float kqi = (float)A.I.ke * (float)pi.w, si = (float)lji.x, ei24 = 24.f * (float)lji.y;
                    v2f fxy = {0.f, 0.f}; float fzs = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t ew = h ? e4.y : e4.x;
                        uint32_t sa = ew & 0x7fffu, sb = (ew >> 16) & 0x7fffu;
                        [[maybe_unused]] const bool real_a = SEG ? (sa - (uint32_t)seg_lo) < (uint32_t)n_here : sa < (uint32_t)tile_n, real_b = SEG ? (sb - (uint32_t)seg_lo) < (uint32_t)n_here : sb < (uint32_t)tile_n;
                        if constexpr (SEG) { sa = real_a ? sa - (uint32_t)seg_lo : (uint32_t)n_here; sb = real_b ? sb - (uint32_t)seg_lo : (uint32_t)n_here; }   // slots of other segments: the sentinel
                        float4 pa, pb; float2 la, lb;
                        pa = *reinterpret_cast<const float4*>(&l_pos[sa]); pb = *reinterpret_cast<const float4*>(&l_pos[sb]);
                        la = *reinterpret_cast<const float2*>(&l_lj[sa]); lb = *reinterpret_cast<const float2*>(&l_lj[sb]);
                        v2f da, db; float dza, dzb;
                        if constexpr (MINIMG) {
                            float x, y, z;
                            min_image_exact<float>((float)pi.x, (float)pi.y, (float)pi.z, pa.x, pa.y, pa.z, reinterpret_cast<const GridP<float>&>(G), x, y, z); da = (v2f){x, y}; dza = z;
                            min_image_exact<float>((float)pi.x, (float)pi.y, (float)pi.z, pb.x, pb.y, pb.z, reinterpret_cast<const GridP<float>&>(G), x, y, z); db = (v2f){x, y}; dzb = z;
                        } else {
                            const v2f pixy = {(float)pi.x, (float)pi.y};
                            da = (v2f){pa.x, pa.y} - pixy; db = (v2f){pb.x, pb.y} - pixy; dza = pa.z - (float)pi.z; dzb = pb.z - (float)pi.z;
                        }
                        const v2f qa = da * da, qb = db * db;
                        v2f r2 = {__builtin_fmaf(dza, dza, qa.x + qa.y), __builtin_fmaf(dzb, dzb, qb.x + qb.y)};
                        // (1e12, not 1e30: pair_eval2's reaction-field branch forms r³ = r²·(r²/r), and 1e30·1e15 is +inf — inf·0 from the sentinel's
                        // zero charge was a NaN in every padded row; r² = 1e12 is beyond any cutoff and keeps r³, the erfc argument and 1/r² finite)
                        if constexpr (MINIMG) { if (G.triclinic) { if (!real_a) r2.x = 1.0e12f; if (!real_b) r2.y = 1.0e12f; } }
                        if constexpr (PRUNE) {
                            if (valid && real_a && r2.x <= (float)A.r_prune2) emit(make_entry((uint32_t)l_new[sa], 0u, 0));
                            if (valid && real_b && r2.y <= (float)A.r_prune2) emit(make_entry((uint32_t)l_new[sb], 0u, 0));
                        }
                        const v2f fr = pair_eval2<COULM>(reinterpret_cast<const InterP<float>&>(A.I), K, r2, kqi, (v2f){pa.w, pb.w}, si, (v2f){la.x, lb.x}, ei24, (v2f){la.y, lb.y});
                        fxy += da * (v2f){fr.x, fr.x}; fxy += db * (v2f){fr.y, fr.y};
                        fzs = __builtin_fmaf(dza, fr.x, __builtin_fmaf(dzb, fr.y, fzs));
                    }
                    fx -= (T)fxy.x; fy -= (T)fxy.y; fz -= (T)fzs;   // force on i is −f (force.jl:873)
                }
            };
            uint2 e_next = (0 < rows) ? my_rows[0] : make_uint2(0, 0);
            // (the fp64 Ewald loop with the in-loop minimum image — 27-image search included — is not unrolled: four copies of it
            // exceed the 256 VGPRs of a 512-lane block and spill)
            constexpr int UNROLL = (sizeof(T) == 8 && (COULM == MHIP_COUL_EWALD_DIRECT || COULM == COUL_EWALD_EXACT)) ? (MINIMG ? 1 : (LJM == LJ_GENERIC && ENERGY ? 2 : 4)) : 4;
            for (int r = 0; r < rows; ++r) {
                const uint2 e4 = e_next;
                if (r + 1 < rows) e_next = my_rows[(int64_t)(r + 1) * A.BI];
                if constexpr (PK2) {
                    if (pk2_ok && (!SPEC || __builtin_amdgcn_ballot_w64(((e4.x | e4.y) & 0x80008000u) != 0u) == 0ull)) { row_pk2(e4); continue; }
                }
#pragma unroll UNROLL
                for (int k = 0; k < 4; ++k) {
                    uint32_t e = ((k < 2 ? e4.x : e4.y) >> (16 * (k & 1))) & 0xffffu;
                    uint32_t slot = entry_slot(e, esh);
                    [[maybe_unused]] const bool real = SEG ? (slot - (uint32_t)seg_lo) < (uint32_t)n_here : slot < (uint32_t)tile_n;   // not a sentinel / other segment
                    if constexpr (SEG) { slot -= (uint32_t)seg_lo; slot = slot < (uint32_t)n_here ? slot : (uint32_t)n_here; }
                    const bool special = SPEC ? entry_special(e, esh) != 0 : false;
                    T4 pj = l_pos[slot];
                    T2 ljj = make2<T>(T(0), T(0));
                    if constexpr (PER_ATOM_LJ) ljj = l_lj[slot];
                    T dx, dy, dz;
                    if constexpr (MINIMG) {
                        min_image_exact<T>(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, G, dx, dy, dz);
                    } else { dx = pj.x - pi.x; dy = pj.y - pi.y; dz = pj.z - pi.z; }
                    T r2 = dx * dx + dy * dy + dz * dz;
                    // the triclinic minimum image folds ANY separation back into the cell, the far-away sentinel atom included
                    if constexpr (MINIMG) { if (G.triclinic && !real) r2 = T(1.0e30); }
                    if constexpr (PRUNE) { if (real && valid && r2 <= A.r_prune2) emit(make_entry((uint32_t)l_new[slot], entry_special(e, esh), esh)); }
                    T fr = pair_eval<T, LJM, COULM, ENERGY, PRE_E>(A.I, r2, pi.w, pj.w, lji.x, ljj.x, lji.y, ljj.y, special, pe);
                    fx -= fr * dx; fy -= fr * dy; fz -= fr * dz;   // force on i is −f (force.jl:873)
                    if constexpr (ENERGY) {
                        vir[0] += fr * dx * dx; vir[1] += fr * dy * dy; vir[2] += fr * dz * dz;
                        vir[3] += fr * dx * dy; vir[4] += fr * dx * dz; vir[5] += fr * dy * dz;
                    }
                }
            }
        };
        // one-type LJ fluids normally have no special pairs at all: a second copy of the loop without the weight selects
        if constexpr (LJM == LJ_DIST_UNIFORM && COULM == MHIP_COUL_NONE && !ENERGY) {
            if (A.any_special) walk_rows(std::true_type{}); else walk_rows(std::false_type{});
        } else walk_rows(std::true_type{});
        if constexpr (!SEG) break;      // (one piece: said outright, or the compiler keeps a loop — and copies of everything that lives across it — around the whole pass)
    }
    if constexpr (PRUNE) {
        // finish the inner list: pad to the wave's row count; record how far the block's atoms moved since the outer build
        const uint32_t SENTP = make_entry((uint32_t)n_new, 0u, esh);
        const int NWB = A.BI >> 6;
        // Two sub-lists per atom (the 256 × 2 shape of the large fluids): which of an atom's kept entries fall to which is a coin toss per
        // entry (tile slot parity), a wave walks as many rows as its longest lane, and 17 % of the slots of a 1M-atom pass were padding.
        // Before the lists are padded the two lanes of an atom level their counts: the longer one hands its LAST entries to the shorter
        // one's tail (2-byte stores into rows both have flushed; the positions written are the receiver's alone, the padding below comes
        // after a barrier).  Every wave of a pass sees the whole tile, so which sub-list holds an entry changes nothing but the order of
        // the atom's sum — fixed, as before — and the pass walks 6 % fewer slots for a few stores per atom and prune (a separate
        // re-dealing pass over the 272 MB list cost what it saved, profiles/r04_force_ab.txt §2).
        const bool level = A.level_pairs && A.JS == 2;
        if (level) {
            if (kept & 3) out_rows[(int64_t)(kept >> 2) * A.BI] = make_uint2((uint32_t)acc, (uint32_t)(acc >> 32));      // my pending entries: the row goes out as it is
            l_cnt[tid] = kept;
            __syncthreads();
            const int kp = l_cnt[tid ^ A.BI], total = kept + kp, mine = js == 0 ? (total + 1) >> 1 : total >> 1;
            if (kept > mine) {
                uint16_t* to = reinterpret_cast<uint16_t*>(A.nbr_dst + (((int64_t)b * A.JS + (js ^ 1)) * A.R_cap) * A.BI + li);
                const uint16_t* from = reinterpret_cast<const uint16_t*>(out_rows);
                for (int q = 0; q < kept - mine; ++q) {
                    const int ps = mine + q, pd = kp + q;
                    to[(((int64_t)(pd >> 2) * A.BI) << 2) + (pd & 3)] = from[(((int64_t)(ps >> 2) * A.BI) << 2) + (ps & 3)];
                }
            }
            kept = mine;
        }
        if (A.cnt_dst) A.cnt_dst[((int64_t)b * A.JS + js) * A.BI + li] = (uint16_t)min(kept, 65535);
        // every lane pads to the row count of its wave
        atomicMax(&l_wmax[js * NWB + (li >> 6)], (kept + 3) >> 2);
        __syncthreads();
        const int rows_wave = l_wmax[js * NWB + (li >> 6)];
        if (level) {      // (the rows are in memory already: the padding goes there too, entry by entry)
            uint16_t* mine16 = reinterpret_cast<uint16_t*>(out_rows);
            for (int p2 = kept; p2 < 4 * rows_wave; ++p2) mine16[(((int64_t)(p2 >> 2) * A.BI) << 2) + (p2 & 3)] = (uint16_t)SENTP;
        } else
        while (((kept + 3) >> 2) < rows_wave || (kept & 3)) emit(SENTP);
        if (tid < A.JS * NWB) A.rows_dst[b * A.JS * NWB + tid] = l_wmax[tid];
        if (tid == 0) A.tile_cnt_dst[b] = n_new;
        float d2 = 0.f;
        if (valid && js == 0) {
            T4 q = A.pos_snap[si];
            T ex = pi_raw.x - q.x, ey = pi_raw.y - q.y, ez = pi_raw.z - q.z;
            disp_image(ex, ey, ez, G);
            d2 = (float)(ex * ex + ey * ey + ez * ez);
        }
        d2 = wave_max(d2);
        if (js == 0 && (tid & 63) == 0) A.blk_disp2[b * (A.BI >> 6) + (li >> 6)] = d2;   // one word per wave of i-atoms: plain stores, nothing to zero beforehand
    }
    stamp(2);
    [[maybe_unused]] T4 st_v, st_p;
    [[maybe_unused]] int64_t se = 0;
    [[maybe_unused]] unsigned long long st_w[3] = {0, 0, 0};
    [[maybe_unused]] int snd0 = 0, snd1 = 0;
    if constexpr (STEP) {      // (in flight across the reduction below: the records, and v_cm as the head workgroup published it — three words, ONE round trip)
        // From here on a wave is a chain of dependent instructions and waits — ≈ 150 of them, which take their turn with the row walks of the seven other
        // waves of the SIMD — while its block keeps a quarter of the compute unit occupied: the arbiter is told to take these waves first.
        __builtin_amdgcn_s_setprio(3);
        se = step_atom();
        if constexpr (HALO) { if ((hflags & 2) && js == 0 && se < A.n_owned) { snd0 = A.H.snd_start[se]; snd1 = A.H.snd_start[se + 1]; } }
        if (js == 0 && se < A.n_owned) {
            st_v = A.vel[se]; st_p = A.pos[se];
            if (A.cm_in) {
#pragma unroll
                for (int c = 0; c < 3; ++c) st_w[c] = __hip_atomic_load(&A.cm_pub[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (A.JS > 1) {   // deterministic reduction of the j-split partial sums through LDS
        __syncthreads();
        T* red = reinterpret_cast<T*>(smem);
        red[(js * 4 + 0) * A.BI + li] = fx; red[(js * 4 + 1) * A.BI + li] = fy;
        red[(js * 4 + 2) * A.BI + li] = fz; red[(js * 4 + 3) * A.BI + li] = pe;
        __syncthreads();
        if (js == 0) {
            for (int q = 1; q < A.JS; ++q) {
                fx += red[(q * 4 + 0) * A.BI + li]; fy += red[(q * 4 + 1) * A.BI + li];
                fz += red[(q * 4 + 2) * A.BI + li]; pe += red[(q * 4 + 3) * A.BI + li];
            }
        }
    }
    if constexpr (STEP) {
        // velocity Verlet for the block's own atoms (simulators.jl:594-616): what k_vv_mid does, with the force still in registers.  Position and velocity
        // are fetched HERE (kept across the pair loop they would cost it its registers); the new position goes to the other buffer.
        double px = 0, py = 0, pz = 0, pm = 0;
        float tr_a = 0.f, tr_b = 0.f, tr_v = 0.f;
        T4 v = st_v, p = st_p;
        const bool mine = js == 0 && se < A.n_owned;
        if (mine) {
            if (A.cm_in) {      // remove_CM_motion! of the step before, one launch late (k_vv_mid's scheme): v_cm as the head workgroup published it
                // (the head workgroup was the first of the grid and finished ≈ 20 µs ago: the words read above are this launch's; if not, again — all three at once)
                if constexpr (HALO) {
                    // (here the head workgroup publishes only once every PEER's sums of the step before have arrived: the wait can be a remote rank's — no
                    // priority over the waves that still compute, a sleep between looks, and an end: the head gives up after the exchange's time-out and publishes anyway)
                    if ((uint32_t)(st_w[0] >> 32) != A.step_seq || (uint32_t)(st_w[1] >> 32) != A.step_seq || (uint32_t)(st_w[2] >> 32) != A.step_seq) {
                        __builtin_amdgcn_s_setprio(0);
                        const unsigned long long t0 = wall_clock64();
                        do {
                            __builtin_amdgcn_s_sleep(16);
#pragma unroll
                            for (int c = 0; c < 3; ++c) st_w[c] = __hip_atomic_load(&A.cm_pub[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } while (((uint32_t)(st_w[0] >> 32) != A.step_seq || (uint32_t)(st_w[1] >> 32) != A.step_seq || (uint32_t)(st_w[2] >> 32) != A.step_seq) && wall_clock64() - t0 < 2 * A.H.ticks);
                        __builtin_amdgcn_s_setprio(3);
                    }
                } else
                while ((uint32_t)(st_w[0] >> 32) != A.step_seq || (uint32_t)(st_w[1] >> 32) != A.step_seq || (uint32_t)(st_w[2] >> 32) != A.step_seq) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) st_w[c] = __hip_atomic_load(&A.cm_pub[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                T vc[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) vc[c] = (T)__uint_as_float((uint32_t)st_w[c]);
                v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2];
                if constexpr (!LANG) { p.x = M<T>::sub(p.x, M<T>::mul(vc[0], A.dt)); p.y = M<T>::sub(p.y, M<T>::mul(vc[1], A.dt)); p.z = M<T>::sub(p.z, M<T>::mul(vc[2], A.dt)); }      // (Langevin: the removal sits behind the step's drifts, simulators.jl:1203 — the velocity alone)
            }
            if constexpr (LANG) {      // the Langevin-middle update (simulators.jl:1171-1201) — the function k_langevin calls — with the force still in registers
                const T4 f4 = make4<T>(fx, fy, fz, T(0));
                langevin_atom<T>(v, p, f4, A.S, (uint64_t)A.orig[se] + 1, G);
                if (A.cm_out) { px = (double)v.x * v.w; py = (double)v.y * v.w; pz = (double)v.z * v.w; pm = v.w; }
            } else {
            const T kx = M<T>::mul(accel_of(fx, v.w), A.dt2), ky = M<T>::mul(accel_of(fy, v.w), A.dt2), kz = M<T>::mul(accel_of(fz, v.w), A.dt2);
            v.x = M<T>::add(v.x, kx); v.y = M<T>::add(v.y, ky); v.z = M<T>::add(v.z, kz);   // :616, v_n before this step's CM removal
            if (A.cm_out) { px = (double)v.x * v.w; py = (double)v.y * v.w; pz = (double)v.z * v.w; pm = v.w; }
            v.x = M<T>::add(v.x, kx); v.y = M<T>::add(v.y, ky); v.z = M<T>::add(v.z, kz);   // :594 of the next step
            p.x = step_add(p.x, v.x, A.dt); p.y = step_add(p.y, v.y, A.dt); p.z = step_add(p.z, v.z, A.dt);   // :602
            wrap_point(p.x, p.y, p.z, G);                                      // :609
            }
            if (A.trk_part) {
                tr_v = (float)(v.x * v.x + v.y * v.y + v.z * v.z);
                auto q = A.snap_a[se];
                T ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
                disp_image(ex, ey, ez, G);
                tr_a = (float)(ex * ex + ey * ey + ez * ez);
                q = A.snap_b[se];
                ex = p.x - q.x; ey = p.y - q.y; ez = p.z - q.z;
                disp_image(ex, ey, ez, G);
                tr_b = (float)(ex * ex + ey * ey + ez * ez);
            }
        }
        if (A.cm_out || A.trk_part) {      // per-block sums / maxima: the waves of the i-atoms (js == 0) by rows, then through LDS, in the fixed order of the integrator kernels
            // (BEHIND the words of the j-split reduction, which slower waves may still be reading: with a j-split its barriers also mean every wave is done with the
            // tile these stores overwrite — a block of several i-waves WITHOUT a j-split has had no barrier since the staging: one here, or a fast wave's sums land in
            // tile coordinates a slower wave is still gathering)
            if (A.JS == 1 && A.BI > 64) __syncthreads();
            double* shd = reinterpret_cast<double*>(smem + (((size_t)A.JS * 4 * A.BI * sizeof(T) + 15) & ~(size_t)15));          // [waves][4 rows][4] doubles, then [waves][4 rows][4] floats
            const int le = step_lane(), nw = A.BI >> 6, w = le >> 6, ln = le & 63;
            float* shf = reinterpret_cast<float*>(shd + 16 * nw);
            if (js == 0) {      // (wave-uniform)
                if (A.cm_out) { const double sv = row_sum4(px, py, pz, pm, ln); if ((ln & 15) < 4) shd[(w * 4 + (ln >> 4)) * 4 + (ln & 3)] = sv; }
                if (A.trk_part) { const float mv = row_max3(tr_a, tr_b, tr_v, ln); if ((ln & 15) < 3) shf[(w * 4 + (ln >> 4)) * 4 + (ln & 3)] = mv; }
            }
            __syncthreads();
            if (A.cm_out && tid < 4) {
                double a = 0;
                for (int q = 0; q < nw; ++q) { const double* r4 = shd + 16 * q + tid; a += (r4[0] + r4[4]) + (r4[8] + r4[12]); }
                if constexpr (HALO) __hip_atomic_store(&A.cm_out[4 * (int64_t)b + tid], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (read by the launch's last wave, on whatever XCD)
                else A.cm_out[4 * (int64_t)b + tid] = a;
            }
            if (A.trk_part && tid < 3) { float mm = 0.f; for (int q = 0; q < 4 * nw; ++q) mm = fmaxf(mm, shf[4 * q + tid]); A.trk_part[(int64_t)tid * A.n_blocks + b] = mm; }
        }
        // (the records go out LAST: a barrier behind a store waits until the store has been acknowledged, and the block would hold its place on the compute unit for that long)
        if (mine) { A.vel[se] = v; A.pos_next[se] = p; }
        if constexpr (HALO) {
            if (js == 0) {      // (wave-uniform)
                if (hflags & 2) {
                    // A block WITHOUT ghosts has not looked at the peers' words yet: exchange seq_send goes into the half the peer read exchange seq_send − 2 from, and
                    // its word for seq_wait = seq_send − 1 says it is done with that (it was sent behind the peer's last read).  Per wave, no barrier.
                    if (!(hflags & 1)) { const int ln = tid & 63; if (ln < A.H.n_peers && !xfer_wait(&A.H.seq_in[A.H.peers[ln]], A.H.seq_wait, A.H.err, A.H.ticks, (9 << 8) | A.H.peers[ln])) atomicOr(A.H.err, 1); }
                    if (mine) for (int r = snd0; r < snd1; ++r) {
                        const HaloSend e = A.H.snd[r];
                        float* d = e.dst + (int64_t)A.H.parity_send * A.H.half_stride;
                        __hip_atomic_store(d, (float)p.x + e.sx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(d + 1, (float)p.y + e.sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(d + 2, (float)p.z + e.sz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                // this wave's stores — rows at the peers, records, the block's partial — are written through and acknowledged before it counts itself in:
                // the wave that finds itself last knows every row of the exchange is in place
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned int before = 0;
                if ((tid & 63) == 0) before = __hip_atomic_fetch_add(A.H.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                before = (unsigned int)__builtin_amdgcn_readfirstlane((int)before);
                if (before == A.H.n_done - 1u) halo_tail(A.H, A.cm_out, A.n_blocks);
            }
        }
    } else {
        if (js == 0 && valid) A.frc[si] = make4<T>(fx, fy, fz, T(0));
    }
    stamp(3);
    if constexpr (!ENERGY) {
        if (A.cm_fin_out && wg == 0) { __syncthreads(); cm_finalize_in_block(A.cm_fin_in, A.cm_fin_n, A.cm_fin_out, smem); }
    }
    if constexpr (ENERGY) {
        if (A.JS > 1) {   // j-split partial sums of the virial, two rounds through the same four LDS slots
            T* red = reinterpret_cast<T*>(smem);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 3; ++c) red[(js * 4 + c) * A.BI + li] = vir[3 * half + c];
                __syncthreads();
                if (js == 0) for (int q = 1; q < A.JS; ++q) for (int c = 0; c < 3; ++c) vir[3 * half + c] += red[(q * 4 + c) * A.BI + li];
            }
        }
        // block sums of the energy and the six virial components, each halved (every pair is visited from both ends);
        // pe_part is component-major: [0, n_blocks) energy, then xx, yy, zz, xy, xz, yz
        __syncthreads();
        double* dred = reinterpret_cast<double*>(smem);
        for (int c = 0; c < 7; ++c) {
            if (js == 0) dred[li] = valid ? 0.5 * (double)(c == 0 ? pe : vir[c - 1]) : 0.0;
            __syncthreads();
            if (tid == 0) { double s = 0; for (int q = 0; q < A.BI; ++q) s += dred[q]; A.pe_part[(int64_t)c * A.n_blocks + b] = s; }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// neighbour list export: half list i<j in caller indices
template <class T>
__global__ void k_export_nl(int n_blocks, int BI, int JS, int T_cap, int R_cap, int64_t n_owned, const int32_t* __restrict__ orig,
                            const int32_t* __restrict__ tile_idx, const int32_t* __restrict__ tile_cnt, const uint2* __restrict__ nbr,
                            const int32_t* __restrict__ wave_rows, int32_t* out_i, int32_t* out_j, uint8_t* out_sp,
                            unsigned long long* counter, unsigned long long capacity, int eshift) {
    int b = blockIdx.x, li = threadIdx.x;
    int64_t si = (int64_t)b * BI + li;
    if (si >= n_owned) return;
    int tile_n = tile_cnt[b];
    int oi = orig[si];
    for (int js = 0; js < JS; ++js) {
        int rows = wave_rows[(b * JS + js) * (BI >> 6) + (li >> 6)];
        for (int r = 0; r < rows; ++r) {
            uint2 e4 = nbr[(((int64_t)b * JS + js) * R_cap + r) * BI + li];
            for (int k = 0; k < 4; ++k) {
                uint32_t e = ((k < 2 ? e4.x : e4.y) >> (16 * (k & 1))) & 0xffffu;
                int slot = (int)entry_slot(e, eshift);
                if (slot >= tile_n) continue;
                int oj = orig[tile_idx[(int64_t)b * T_cap + slot]];
                if (oi < oj) {
                    unsigned long long at = atomicAdd(counter, 1ull);
                    if (out_i && at < capacity) { out_i[at] = oi; out_j[at] = oj; out_sp[at] = (uint8_t)entry_special(e, eshift); }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// per block: does its tile hold a ghost atom?  (Ghosts sort after all owned atoms.)  One wave per block.
[[maybe_unused]] static __global__ void k_block_ghost_flags(int n_blocks, int T_cap, int64_t n_owned, const int32_t* __restrict__ tile_idx, const int32_t* __restrict__ tile_cnt, int32_t* out) {
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    bool g = false;
    const int n = tile_cnt[b];
    for (int t = threadIdx.x; t < n; t += blockDim.x) g = g || tile_idx[(int64_t)b * T_cap + t] >= n_owned;
    const unsigned long long any = __ballot(g);
    if (threadIdx.x == 0) out[b] = any != 0ull;
}

// ---------------------------------------------------------------------------------------------------
// velocity Verlet (simulators.jl:589-629), owned atoms, sorted order.  vel.w carries the mass.
// v_cm = Σ(m v) / Σ m from the per-block partials of k_vv2, re-summed by EVERY block in the same fixed order (n_part <= 1024,
// 32 KB of L2 reads per block): no separate finalize launch between the second kick and the next first kick.
template <class T>
__device__ inline void block_vcm(const double* __restrict__ part, int n_part, T* vcm3) {
    __shared__ double sh_cm[4][4];
    double a[4] = {0, 0, 0, 0};
    for (int q = threadIdx.x; q < n_part; q += blockDim.x) { const double* p = part + 4 * (int64_t)q; a[0] += p[0]; a[1] += p[1]; a[2] += p[2]; a[3] += p[3]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], o, 64);
    if ((threadIdx.x & 63) == 0) for (int c = 0; c < 4; ++c) sh_cm[threadIdx.x >> 6][c] = a[c];
    __syncthreads();
    double t[4] = {0, 0, 0, 0};
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) for (int c = 0; c < 4; ++c) t[c] += sh_cm[q][c];
    for (int c = 0; c < 3; ++c) vcm3[c] = (T)(t[c] / t[3]);
}

// The integrator's arithmetic is the reference's, operation by operation (no fused multiply-add, a true division): a = f / m with 0 for
// massless atoms (calc_accels, force.jl:17), v += a·dt/2 (simulators.jl:594, 616), x += v·dt (:602).  Every kernel below goes through
// these two helpers, so a run cut into chunks repeats the uncut run bit for bit (test/simulation.jl:16-57).
// (accel_of / step_add are defined in front of k_forces, whose STEP variants integrate in their epilogue)

template <class T>
__global__ void k_vv1(int64_t n, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel, const typename Vec<T>::T4* __restrict__ frc,
                      T dt, T dt2, const T* __restrict__ vcm, const double* __restrict__ cm_part, int n_cm_part, GridP<T> G) {
    T vc[3] = {T(0), T(0), T(0)};
    const bool sub = vcm != nullptr || cm_part != nullptr;
    if (cm_part) block_vcm<T>(cm_part, n_cm_part, vc);
    else if (vcm) { vc[0] = vcm[0]; vc[1] = vcm[1]; vc[2] = vcm[2]; }
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto v = vel[s]; auto p = pos[s]; auto f = frc[s];
        if (sub) { v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2]; }                // deferred remove_CM_motion!
        v.x = step_add(v.x, accel_of(f.x, v.w), dt2); v.y = step_add(v.y, accel_of(f.y, v.w), dt2); v.z = step_add(v.z, accel_of(f.z, v.w), dt2);   // :594
        p.x = step_add(p.x, v.x, dt); p.y = step_add(p.y, v.y, dt); p.z = step_add(p.z, v.z, dt);   // :602
        wrap_point(p.x, p.y, p.z, G);                                          // :609
        vel[s] = v; pos[s] = p;
    }
}

// fa (nullable): a second force array — the bonded sums + reciprocal-space forces of a small system's fused launches (step_fused.h) —
// folded in here; the total is written back so that the next first kick sees it
template <class T, bool CM>
__global__ void k_vv2(int64_t n, typename Vec<T>::T4* vel, typename Vec<T>::T4* frc, T dt2, double* cm_part,
                      const typename Vec<T>::T4* __restrict__ fa) {
    double px = 0, py = 0, pz = 0, m = 0;
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto v = vel[s]; auto f = frc[s];
        if (fa) { const auto g = fa[s]; f.x += g.x; f.y += g.y; f.z += g.z; }
        if (fa) frc[s] = f;
        v.x = step_add(v.x, accel_of(f.x, v.w), dt2); v.y = step_add(v.y, accel_of(f.y, v.w), dt2); v.z = step_add(v.z, accel_of(f.z, v.w), dt2);   // :616
        vel[s] = v;
        if constexpr (CM) { px += (double)v.x * v.w; py += (double)v.y * v.w; pz += (double)v.z * v.w; m += v.w; }
    }
    if constexpr (CM) {
        __shared__ double sh[4][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { px += __shfl_xor(px, o, 64); py += __shfl_xor(py, o, 64); pz += __shfl_xor(pz, o, 64); m += __shfl_xor(m, o, 64); }
        int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh[w][0] = px; sh[w][1] = py; sh[w][2] = pz; sh[w][3] = m; }
        __syncthreads();
        if (threadIdx.x < 4) { double a = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) a += sh[q][threadIdx.x]; cm_part[4 * (int64_t)blockIdx.x + threadIdx.x] = a; }
    }
}

// Second kick of step n and first kick + drift of step n+1 in ONE pass over the atoms (vv_run, single domain): halves the integrator's
// launches and its HBM traffic (48 B read + 32 B written per fp32 atom instead of 80 + 48).  remove_CM_motion! sits between the two
// kicks and needs Σ m v of ALL atoms, so it is applied one kernel late: this launch accumulates the partials of Σ m v_n (before the
// removal), carries on with the unshifted velocity, and the NEXT launch subtracts v_cm from the velocity it finds and v_cm·dt from
// the position that was drifted with it.  The forces in between saw every atom translated by the same v_cm·dt (≈ 1e-11 nm): they are
// translation invariant.  LAST: stop after the second kick (the run's final step), leaving v_n and x_n for the caller.
template <class T, bool CM, bool LAST>
__global__ void k_vv_mid(int64_t n, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel, const typename Vec<T>::T4* __restrict__ fr, T dt, T dt2,
                         const double* __restrict__ cm_in, int n_cm_in, double* cm_out,
                         const typename Vec<T>::T4* __restrict__ fa, GridP<T> G,
                         const typename Vec<T>::T4* __restrict__ snap_a = nullptr, const typename Vec<T>::T4* __restrict__ snap_b = nullptr, float* trk_part = nullptr) {
    const typename Vec<T>::T4* __restrict__ frc = fr;
    // trk_part (the validity check of the pair lists, taken where the new coordinates are made): per block the largest |x − snap_a|²,
    // |x − snap_b|² and |v|² of what this launch leaves behind, trk_part[c·gridDim.x + block]
    float v2m = 0.f, dam = 0.f, dbm = 0.f;
    T vc[3] = {T(0), T(0), T(0)};
    const bool sub = cm_in != nullptr;
    // A lane's first atom is requested BEFORE the centre-of-mass partials are re-summed (32 KB of L2 reads, two barriers): the
    // streaming part of the launch then starts behind that latency instead of after it.
    using T4q = typename Vec<T>::T4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, sn = s;
    T4q vq = make4<T>(T(0), T(0), T(0), T(1)), fq = vq, pq = vq, gaq = vq;
    auto fetch = [&](int64_t a) {
        vq = vel[a]; fq = frc[a];
        if (!LAST || sub) pq = pos[a];
        if (fa) gaq = fa[a];
    };
    if (s < n) fetch(s);
    if (sub) block_vcm<T>(cm_in, n_cm_in, vc);
    const T sh[3] = {M<T>::mul(vc[0], dt), M<T>::mul(vc[1], dt), M<T>::mul(vc[2], dt)};
    double px = 0, py = 0, pz = 0, m = 0;
    for (; s < n; s = sn) {
        sn = s + stride;
        const auto v0 = vq, f0 = fq, p0 = pq, ga0 = gaq;
        if (sn < n) fetch(sn);                                                 // the next atom's data travel while this one is integrated
        auto v = v0; auto f = f0; auto p = p0;
        if (fa) { f.x += ga0.x; f.y += ga0.y; f.z += ga0.z; }
        if (sub) {                                                             // remove_CM_motion! of the previous step, one launch late
            v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2];
            p.x = M<T>::sub(p.x, sh[0]); p.y = M<T>::sub(p.y, sh[1]); p.z = M<T>::sub(p.z, sh[2]);
        }
        const T kx = M<T>::mul(accel_of(f.x, v.w), dt2), ky = M<T>::mul(accel_of(f.y, v.w), dt2), kz = M<T>::mul(accel_of(f.z, v.w), dt2);
        v.x = M<T>::add(v.x, kx); v.y = M<T>::add(v.y, ky); v.z = M<T>::add(v.z, kz);   // :616, v_n before this step's CM removal
        if constexpr (CM) { px += (double)v.x * v.w; py += (double)v.y * v.w; pz += (double)v.z * v.w; m += v.w; }
        if constexpr (!LAST) {
            v.x = M<T>::add(v.x, kx); v.y = M<T>::add(v.y, ky); v.z = M<T>::add(v.z, kz);   // :594 of the next step
            p.x = step_add(p.x, v.x, dt); p.y = step_add(p.y, v.y, dt); p.z = step_add(p.z, v.z, dt);   // :602
        }
        if (!LAST || sub) {
            wrap_point(p.x, p.y, p.z, G);                                      // :609
            pos[s] = p;
        }
        if (LAST && fa) const_cast<typename Vec<T>::T4*>(frc)[s] = f;  // the total force of the last step stays readable
        vel[s] = v;
        if (trk_part) {
            v2m = fmaxf(v2m, (float)(v.x * v.x + v.y * v.y + v.z * v.z));
            auto q = snap_a[s];
            T ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
            disp_image(ex, ey, ez, G);
            dam = fmaxf(dam, (float)(ex * ex + ey * ey + ez * ez));
            q = snap_b[s];
            ex = p.x - q.x; ey = p.y - q.y; ez = p.z - q.z;
            disp_image(ex, ey, ez, G);
            dbm = fmaxf(dbm, (float)(ex * ex + ey * ey + ez * ez));
        }
    }
    if (trk_part) {
        __shared__ float sht[3][16];
        dam = wave_max(dam); dbm = wave_max(dbm); v2m = wave_max(v2m);
        if ((threadIdx.x & 63) == 0) { sht[0][threadIdx.x >> 6] = dam; sht[1][threadIdx.x >> 6] = dbm; sht[2][threadIdx.x >> 6] = v2m; }
        __syncthreads();
        if (threadIdx.x < 3) { float m = 0.f; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) m = fmaxf(m, sht[threadIdx.x][q]); trk_part[threadIdx.x * gridDim.x + blockIdx.x] = m; }
    }
    if constexpr (CM) {
        __shared__ double shm[4][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { px += __shfl_xor(px, o, 64); py += __shfl_xor(py, o, 64); pz += __shfl_xor(pz, o, 64); m += __shfl_xor(m, o, 64); }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { shm[w][0] = px; shm[w][1] = py; shm[w][2] = pz; shm[w][3] = m; }
        __syncthreads();
        if (threadIdx.x < 4) { double a = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) a += shm[q][threadIdx.x]; cm_out[4 * (int64_t)blockIdx.x + threadIdx.x] = a; }
    }
}

// frc += fa: the same fold outside the integrator
template <class T>
__global__ void k_add_forces(int64_t n, typename Vec<T>::T4* frc, const typename Vec<T>::T4* __restrict__ fa) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto f = frc[s];
        if (fa) { const auto g = fa[s]; f.x += g.x; f.y += g.y; f.z += g.z; }
        frc[s] = f;
    }
}

// Σ m v and Σ m of the current velocities (for an explicit remove_CM_motion! / multi-GPU all-reduce)
template <class T>
__global__ void k_cm_partials(int64_t n, const typename Vec<T>::T4* __restrict__ vel, const T* __restrict__ vcm, double* cm_part) {
    int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double px = 0, py = 0, pz = 0, m = 0;
    if (s < n) { auto v = vel[s]; if (vcm) { v.x -= vcm[0]; v.y -= vcm[1]; v.z -= vcm[2]; } px = (double)v.x * v.w; py = (double)v.y * v.w; pz = (double)v.z * v.w; m = v.w; }
    __shared__ double sh[4][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { px += __shfl_xor(px, o, 64); py += __shfl_xor(py, o, 64); pz += __shfl_xor(pz, o, 64); m += __shfl_xor(m, o, 64); }
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = px; sh[w][1] = py; sh[w][2] = pz; sh[w][3] = m; }
    __syncthreads();
    if (threadIdx.x < 4) { double a = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) a += sh[q][threadIdx.x]; cm_part[4 * (int64_t)blockIdx.x + threadIdx.x] = a; }
}

// one block of 256 threads: fixed-order sum of the per-block partials → out4 = {Px,Py,Pz,M} (double), vcm = P/M (T)
template <class T>
__global__ void k_cm_finalize(int n_part, const double* __restrict__ cm_part, double* out4, T* vcm) {
    __shared__ double sh[4][4];
    double a[4] = {0, 0, 0, 0};
    for (int q = threadIdx.x; q < n_part; q += blockDim.x) { const double* p = cm_part + 4 * (int64_t)q; a[0] += p[0]; a[1] += p[1]; a[2] += p[2]; a[3] += p[3]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], o, 64);
    if ((threadIdx.x & 63) == 0) for (int c = 0; c < 4; ++c) sh[threadIdx.x >> 6][c] = a[c];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4] = {0, 0, 0, 0};
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) for (int c = 0; c < 4; ++c) t[c] += sh[q][c];
        for (int c = 0; c < 4; ++c) out4[c] = t[c];
        if (vcm) for (int c = 0; c < 3; ++c) vcm[c] = (T)(t[c] / t[3]);
    }
}

// vcm = P/M from a device-resident {Px,Py,Pz,M} (the all-reduced total of a multi-GPU run)
template <class T>
__global__ void k_vcm_from_total(const double* __restrict__ total4, T* vcm) {
    if (threadIdx.x < 3) vcm[threadIdx.x] = (T)(total4[threadIdx.x] / total4[3]);
}

template <class T>
__global__ void k_shift_vel(int64_t n, typename Vec<T>::T4* vel, const T* __restrict__ vcm, const double* __restrict__ cm_part, int n_cm_part) {
    T vc[3];
    if (cm_part) block_vcm<T>(cm_part, n_cm_part, vc);
    else { vc[0] = vcm[0]; vc[1] = vcm[1]; vc[2] = vcm[2]; }
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto v = vel[s]; v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2]; vel[s] = v;
    }
}

// kinetic energy partials: Σ (m/2) v·v  (energy.jl:56-89)
template <class T>
__global__ void k_ke_partials(int64_t n, const typename Vec<T>::T4* __restrict__ vel, double* part) {
    int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double k = 0;
    if (s < n) { auto v = vel[s]; k = 0.5 * (double)v.w * ((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z); }
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) { double a = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) a += sh[q]; part[blockIdx.x] = a; }
}

// fixed-order sum of n doubles per block: block c sums part[c·n .. (c+1)·n) into out[c] (component-major partial arrays)
[[maybe_unused]] static __global__ void k_sum_double(int n, const double* __restrict__ part, double* out) {
    __shared__ double sh[256];
    part += (size_t)blockIdx.x * n;
    double a = 0;
    for (int q = threadIdx.x; q < n; q += blockDim.x) a += part[q];
    sh[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int q = 0; q < (int)blockDim.x; ++q) t += sh[q]; out[blockIdx.x] = t; }
}

template <class T>
__global__ void k_check_finite(int64_t n, const typename Vec<T>::T4* __restrict__ pos, const typename Vec<T>::T4* __restrict__ vel,
                               const typename Vec<T>::T4* __restrict__ frc, int32_t* flags) {
    int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= n) return;
    auto p = pos[s]; auto v = vel[s]; auto f = frc[s];
    T t = p.x + p.y + p.z + v.x + v.y + v.z + f.x + f.y + f.z;
    if (!(t == t) || M<T>::fabs(t) > T(1e30)) atomicOr(&flags[FLAG_NAN], 1);
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU halo helpers
template <class T>
__global__ void k_gather_coords(int64_t n, const int32_t* __restrict__ idx, const T* __restrict__ shift, const int32_t* __restrict__ inv,
                                const typename Vec<T>::T4* __restrict__ pos, T* out) {
    int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    auto p = pos[inv[idx[k]]];
    T sx = shift ? shift[3 * k] : T(0), sy = shift ? shift[3 * k + 1] : T(0), sz = shift ? shift[3 * k + 2] : T(0);
    out[3 * k] = p.x + sx; out[3 * k + 1] = p.y + sy; out[3 * k + 2] = p.z + sz;
}
template <class T>
__global__ void k_scatter_coords(int64_t first, int64_t n, const T* __restrict__ in, const int32_t* __restrict__ inv, typename Vec<T>::T4* pos) {
    int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    int s = inv[first + k];
    pos[s].x = in[3 * k]; pos[s].y = in[3 * k + 1]; pos[s].z = in[3 * k + 2];
}


// The per-step halo message of a ghost plan (mhip_halo_plan): coordinate rows of the atoms the peers need and, after each peer's
// rows, cm_rows rows that carry this rank's {ΣPx, ΣPy, ΣPz, ΣM} of the step before (four doubles seen as words of T, three per
// row) — the centre-of-mass sum rides on the ghost exchange instead of an all-reduce of its own.  idx < 0 marks those rows; they are
// written by ONE extra block that re-sums the per-block partials of the integrator launch (cm_part, n_part; nullptr: zeros) and
// also leaves the total in cm_own, the first slot of the table the next integrator launch sums.
// X.row_peer != nullptr: the rows do not go to the local send buffer but straight into the peers' receive regions (halo_xfer.h), and
// the last block to finish raises this rank's sequence word there — packing and sending in one launch.
template <class T>
__global__ void __launch_bounds__(256) k_halo_pack(int64_t n, const int32_t* __restrict__ idx, const T* __restrict__ shift, const int32_t* __restrict__ inv,
                                                   const typename Vec<T>::T4* __restrict__ pos, T* out, const double* __restrict__ cm_part, int n_part,
                                                   const int32_t* __restrict__ cm_pos, int n_cm_pos, int cm_rows, double* cm_own, XferSend X) {
    auto row_ptr = [&](int64_t k) -> T* {
        if (!X.row_peer) return out + 3 * k;
        return reinterpret_cast<T*>(X.P.region[X.row_peer[k]] + XFER_ROWS_OFF) + (size_t)X.parity * X.rows_cap * 3 + 3 * (size_t)X.row_dst[k];
    };
    if (blockIdx.x == gridDim.x - 1) {
        __shared__ double tot[4];
        __shared__ double sh_cm[4][4];
        double a[4] = {0, 0, 0, 0};
        if (cm_part) for (int q = threadIdx.x; q < n_part; q += blockDim.x) { const double* p = cm_part + 4 * (int64_t)q; a[0] += p[0]; a[1] += p[1]; a[2] += p[2]; a[3] += p[3]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], o, 64);
        if ((threadIdx.x & 63) == 0) for (int c = 0; c < 4; ++c) sh_cm[threadIdx.x >> 6][c] = a[c];
        __syncthreads();
        if (threadIdx.x < 4) { double t = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += sh_cm[q][threadIdx.x]; tot[threadIdx.x] = t; if (cm_own) cm_own[threadIdx.x] = t; }
        __syncthreads();
        constexpr int NW = 32 / (int)sizeof(T);
        const T* w = reinterpret_cast<const T*>(tot);
        for (int q = threadIdx.x; q < n_cm_pos; q += blockDim.x) {
            const int r = q % cm_rows; const int64_t k = cm_pos[q];
            T* o = row_ptr(k);
            for (int c = 0; c < 3; ++c) o[c] = (3 * r + c < NW) ? w[3 * r + c] : T(0);
        }
        if (X.row_peer) xfer_announce(X);
        return;
    }
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k < n) {
        const int i = idx[k];
        if (i >= 0) {
            const auto p = pos[inv[i]];
            T* o = row_ptr(k);
            o[0] = p.x + shift[3 * k]; o[1] = p.y + shift[3 * k + 1]; o[2] = p.z + shift[3 * k + 2];
        }
    }
    if (X.row_peer) xfer_announce(X);
}
// the receiving side: dst >= 0 → coordinates of ghost slot first + dst; dst = −1 − (peer·cm_rows + r) → row r of that peer's sums,
// collected in cm_all[1 + peer] (slot 0 is this rank's own, k_halo_pack)
// W.n_peers > 0: the rows come from this rank's receive region; every block first waits for the senders' sequence words (halo_xfer.h)
template <class T>
__global__ void k_halo_unpack(int64_t n, const T* __restrict__ in, const int32_t* __restrict__ dst, int64_t first, const int32_t* __restrict__ inv,
                              typename Vec<T>::T4* pos, double* cm_all, int cm_rows, XferWait W) {
    if (W.n_peers > 0) xfer_wait_block(W);
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int d = dst[k];
    if (d >= 0) { const int s = inv[first + d]; pos[s].x = peer_row_load(in + 3 * k); pos[s].y = peer_row_load(in + 3 * k + 1); pos[s].z = peer_row_load(in + 3 * k + 2); return; }
    constexpr int NW = 32 / (int)sizeof(T);
    const int code = -1 - d, peer = code / cm_rows, r = code - peer * cm_rows;
    T* w = reinterpret_cast<T*>(cm_all + 4 * (int64_t)(1 + peer));
    for (int c = 0; c < 3; ++c) if (3 * r + c < NW) w[3 * r + c] = peer_row_load(in + 3 * k + c);
}

}  // namespace mhip

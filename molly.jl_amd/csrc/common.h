// common.h — shared types for the MI355X (gfx950) nonbonded engine.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/mollyhip.h"

namespace mhip {

constexpr int WAVE = 64;
// The one compile-time experiment switch of the product sources: -DMHIP_STAMPS=1 builds a library whose block kernels (k_forces, k_forces_gs, k_build,
// k_regroup) leave wall-clock stamps per wave for tools/gs_times.py and tools/build_times.py (MOLLYHIP_DBG_TIMES / MOLLYHIP_DBG_DUMP* in the engine).
#ifndef MHIP_STAMPS
#define MHIP_STAMPS 0
#endif
// dynamic LDS a kernel may ask for: the 160 KiB of a gfx950 CU (MI355X_MICROARCH.md) minus room for the kernels' static __shared__
// arrays (k_build: ≈ 1.6 KB) — the sum is what hipFuncSetAttribute / the launch are checked against
constexpr int MAX_LDS_BYTES = 160 * 1024 - 2048;
constexpr int COUL_EWALD_EXACT = 4;        // pair-kernel variant of MHIP_COUL_EWALD_DIRECT with the libm erfc (approximate_erfc = false): a template value, so that
                                           // the default Abramowitz-Stegun form is straight-line code in the loop
constexpr int TILE_SLOT_MAX = 32767;        // 15-bit tile slot + 1-bit special flag per list entry

template <class T> struct Vec;
template <> struct Vec<float> { using T4 = float4; using T2 = float2; };
template <> struct Vec<double> { using T4 = double4; using T2 = double2; };

template <class T> __host__ __device__ inline typename Vec<T>::T4 make4(T x, T y, T z, T w) {
    typename Vec<T>::T4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v;
}
template <class T> __host__ __device__ inline typename Vec<T>::T2 make2(T x, T y) {
    typename Vec<T>::T2 v; v.x = x; v.y = y; return v;
}

// Geometry of the (sub-)domain: orthorhombic box, cell grid of the neighbour search.
template <class T> struct GridP {
    T L[3], invL[3], origin[3];
    T cs[3], inv_cs[3];        // cell size (>= r_list / stencil) and its inverse
    int periodic[3];
    int nc[3];                 // cells per axis
    int stencil[3];            // cells to visit on each side (clamped so a cell is never visited twice)
    int all_cells[3];          // 1: visit every cell of that axis (small boxes / no neighbour list)
    int ncell;
    T r_list, r_list2;         // +inf when every pair interacts
    int no_list;
    // TriclinicBoundary (spatial.jl:131-220): 0 = cubic; 1 = approx_images (three floor steps, :528-534); 2 = the exact search over
    // the 27 neighbouring images (:536-551).  The whole system is then ONE cell: every block's tile holds every atom and all
    // distances take the exact in-loop minimum image (small systems only, as in test/gpu_consistency.jl:287-337).
    int triclinic;
    T bv[3][3];                // basis vectors (rows): a ∥ x, b in the xy plane
    T rs[3];                   // reciprocal_size = 1/a_x, 1/b_y, 1/c_z
    T cot_bc, cxz, cyz, cot_ab;   // wrap_coords constants (:192-210, 588-602)
    // Cell grid of a triclinic box (tri_grid = 1): the cells live in u = s·h, fractional coordinates scaled by the perpendicular
    // heights h of the cell.  u_d is the projection of x on the unit normal of face pair d, so |Δu_d| <= |Δx| on every axis: the grid,
    // the bounding boxes and every pruning step work per axis in u (a Chebyshev test), distances are Cartesian.
    int tri_grid;
    T hgt[3];
};

// pairwise_inters in device-friendly form (constants rounded to T on the host exactly as the
// reference stores them inside the interaction structs)
template <class T> struct InterP {
    int lj, lj_cut; T lj_rc, lj_rc2, lj_ra, lj_w;
    T lj_s2, lj_24e, lj_4e;    // uniform-LJ fast path: σ², 24ϵ, 4ϵ of the single atom type
    T lj_c6, lj_c12;           // … and 24ϵσ⁶, 48ϵσ¹² (the packed loop's F/r = (c12/r⁶ − c6)/r⁸; 0 when they leave the normal fp32 range)
    int coul, coul_cut; T c_rc, c_rc2, c_ra, ke, c_w;
    T krf, crf;                // reaction-field constants for non-special pairs (coulomb.jl:764-768,799-803)
    T alpha, two_over_sqrt_pi;
    int approx_erfc;
};

struct HipErr { hipError_t e; const char* what; };
#define MHIP_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw mhip::HipErr{_e, #expr}; } while (0)

struct ApiError { int32_t code; std::string msg; };

inline int cdiv(int64_t a, int64_t b) { return int((a + b - 1) / b); }

}  // namespace mhip

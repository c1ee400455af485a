// philox.h — the counter-based noise of the stochastic pieces as device functions (see stochastic.h): Philox4x32-10, the uniform → normal transform, three normals per
// atom, and the Langevin-middle update of ONE atom — shared by the stand-alone update kernel (stochastic.hip) and by the last force launch of a small system's step
// when that launch integrates (step_fused.h), so that both run the same arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "common.h"
#include "physics.h"
#include "stochastic.h"

namespace mhip {

struct U4 { uint32_t a, b, c, d; };

// Philox4x32-10: ten rounds of two 32x32→64 multiplies with the Weyl-bumped key (Random123 philox.h, constants as published)
__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.a, p1 = (uint64_t)0xCD9E8D57u * c.c;
        U4 n;
        n.a = (uint32_t)(p1 >> 32) ^ c.b ^ k0; n.b = (uint32_t)p1;
        n.c = (uint32_t)(p0 >> 32) ^ c.d ^ k1; n.d = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
__host__ __device__ inline U4 philox_u64(uint64_t ctr0, uint64_t ctr1, uint64_t key) {
    U4 c; c.a = (uint32_t)ctr0; c.b = (uint32_t)(ctr0 >> 32); c.c = (uint32_t)ctr1; c.d = (uint32_t)(ctr1 >> 32);
    return philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
}

// Box-Muller on uniforms from the OPEN interval (k + ½)·2⁻ᵇ: no log(0), symmetric about ½
__device__ inline void box_muller(float u1, float u2, float& z0, float& z1) {
    const float r = ::sqrtf(-2.0f * ::logf(u1));
    float sn, cs; ::sincosf(6.2831853071795864769f * u2, &sn, &cs);
    z0 = r * cs; z1 = r * sn;
}
__device__ inline void box_muller(double u1, double u2, double& z0, double& z1) {
    const double r = ::sqrt(-2.0 * ::log(u1));
    double sn, cs; ::sincos(6.2831853071795864769 * u2, &sn, &cs);
    z0 = r * cs; z1 = r * sn;
}
__device__ inline float unit_f32(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-8f; }                    // 2⁻²⁴
__device__ inline double unit_f64(uint32_t lo, uint32_t hi) { return ((double)((((uint64_t)hi << 32) | lo) >> 11) + 0.5) * 1.1102230246251565e-16; }   // 2⁻⁵³

// three standard normals for atom ctr0 (≙ randn_svec kernels.jl:664-686): fp32 spends ONE Philox block (four words → four
// normals, the last unused), fp64 two blocks — the second at ctr0 + natoms, which is why callers that draw twice advance by natoms
template <class T> __device__ inline void randn3(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint64_t natoms, T* z);
template <> __device__ inline void randn3<float>(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint64_t, float* z) {
    const U4 w = philox_u64(ctr0, ctr1, key);
    float spare;
    box_muller(unit_f32(w.a), unit_f32(w.b), z[0], z[1]);
    box_muller(unit_f32(w.c), unit_f32(w.d), z[2], spare);
}
template <> __device__ inline void randn3<double>(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint64_t natoms, double* z) {
    const U4 w = philox_u64(ctr0, ctr1, key), w2 = philox_u64(ctr0 + natoms, ctr1, key);
    double spare;
    box_muller(unit_f64(w.a, w.b), unit_f64(w.c, w.d), z[0], z[1]);
    box_muller(unit_f64(w2.a, w2.b), unit_f64(w2.c, w2.d), z[2], spare);
}

template <class T> __device__ inline T fma_t(T a, T b, T c);
template <> __device__ inline float fma_t<float>(float a, float b, float c) { return ::fmaf(a, b, c); }
template <> __device__ inline double fma_t<double>(double a, double b, double c) { return ::fma(a, b, c); }

// sqrt(kT / m) in double, rounded once to T (virtual sites / massless atoms: 0, simulators.jl:1145-1147)
template <class T> __device__ inline T thermal_scale(double noise_kt, T m) { return m == T(0) ? T(0) : (T)(noise_kt * ::sqrt(1.0 / (double)m)); }

// the update of one atom (simulators.jl:1171-1201): v += (f/m)·dt ; x = muladd(dt/2, v, x) ; v = muladd(vel_scale, v, noise·noise_scale) ; x = muladd(dt/2, v, x) ; wrap.
// ctr0: the atom's 1-based ORIGINAL index.  Every operation individually rounded (explicit fma where the reference writes muladd).
template <class T, class T4> __device__ inline void langevin_atom(T4& v, T4& p, const T4& f, const StochP<T>& P, uint64_t ctr0, const GridP<T>& G) {
#pragma clang fp contract(off)
    const T im = (v.w == T(0)) ? T(0) : T(1) / v.w;                        // calc_accels, force.jl:17
    v.x += (f.x * im) * P.dt; v.y += (f.y * im) * P.dt; v.z += (f.z * im) * P.dt;             // :1176
    p.x = fma_t(P.dt_half, v.x, p.x); p.y = fma_t(P.dt_half, v.y, p.y); p.z = fma_t(P.dt_half, v.z, p.z);   // :1187
    T z[3];
    randn3<T>(ctr0, P.ctr1, P.key, P.natoms, z);
    const T ns = thermal_scale<T>(P.noise_kt, v.w);
    v.x = fma_t(P.vel_scale, v.x, z[0] * ns); v.y = fma_t(P.vel_scale, v.y, z[1] * ns); v.z = fma_t(P.vel_scale, v.z, z[2] * ns);   // kernels.jl:739
    p.x = fma_t(P.dt_half, v.x, p.x); p.y = fma_t(P.dt_half, v.y, p.y); p.z = fma_t(P.dt_half, v.z, p.z);   // :1192
    wrap_point(p.x, p.y, p.z, G);                                          // :1201
}

}  // namespace mhip

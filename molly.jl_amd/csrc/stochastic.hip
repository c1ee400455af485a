// stochastic.hip — Philox4x32-10 noise, the fused Langevin update, Andersen / Maxwell-Boltzmann velocity draws (see stochastic.h)
#include "stochastic.h"

#include "kernels.h"

namespace mhip {

namespace {

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

// One Langevin-middle step of the owned atoms, forces already at hand (simulators.jl:1171-1197):
//   v += (f/m)·dt ; x = muladd(dt/2, v, x) ; v = muladd(vel_scale, v, noise·noise_scale) ; x = muladd(dt/2, v, x) ; wrap
// plus the deferred remove_CM_motion! of the previous step in front and the Σ m v partials of this one behind.
template <class T, bool CM>
__global__ void __launch_bounds__(256) k_langevin(int64_t n, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel, const typename Vec<T>::T4* __restrict__ frc,
                                                  const int32_t* __restrict__ orig, StochP<T> P, const T* __restrict__ vcm, const double* __restrict__ cm_in,
                                                  int n_cm_in, double* cm_out, GridP<T> G) {
#pragma clang fp contract(off)
    T vc[3] = {T(0), T(0), T(0)};
    const bool sub = vcm != nullptr || cm_in != nullptr;
    if (cm_in) block_vcm<T>(cm_in, n_cm_in, vc);
    else if (vcm) { vc[0] = vcm[0]; vc[1] = vcm[1]; vc[2] = vcm[2]; }
    double px = 0, py = 0, pz = 0, m = 0;
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto v = vel[s]; auto p = pos[s]; const auto f = frc[s];
        if (sub) { v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2]; }
        const T im = (v.w == T(0)) ? T(0) : T(1) / v.w;                        // calc_accels, force.jl:17
        v.x += (f.x * im) * P.dt; v.y += (f.y * im) * P.dt; v.z += (f.z * im) * P.dt;             // :1176
        p.x = fma_t(P.dt_half, v.x, p.x); p.y = fma_t(P.dt_half, v.y, p.y); p.z = fma_t(P.dt_half, v.z, p.z);   // :1187
        T z[3];
        randn3<T>((uint64_t)orig[s] + 1, P.ctr1, P.key, P.natoms, z);
        const T ns = thermal_scale<T>(P.noise_kt, v.w);
        v.x = fma_t(P.vel_scale, v.x, z[0] * ns); v.y = fma_t(P.vel_scale, v.y, z[1] * ns); v.z = fma_t(P.vel_scale, v.z, z[2] * ns);   // kernels.jl:739
        p.x = fma_t(P.dt_half, v.x, p.x); p.y = fma_t(P.dt_half, v.y, p.y); p.z = fma_t(P.dt_half, v.z, p.z);   // :1192
        wrap_point(p.x, p.y, p.z, G);                                          // :1201
        vel[s] = v; pos[s] = p;
        if constexpr (CM) { px += (double)v.x * v.w; py += (double)v.y * v.w; pz += (double)v.z * v.w; m += v.w; }
    }
    if constexpr (CM) {
        __shared__ double sh[4][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { px += __shfl_xor(px, o, 64); py += __shfl_xor(py, o, 64); pz += __shfl_xor(pz, o, 64); m += __shfl_xor(m, o, 64); }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh[w][0] = px; sh[w][1] = py; sh[w][2] = pz; sh[w][3] = m; }
        __syncthreads();
        if (threadIdx.x < 4) { double a = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) a += sh[q][threadIdx.x]; cm_out[4 * (int64_t)blockIdx.x + threadIdx.x] = a; }
    }
}

// MODE 0 (Andersen): the first two Philox words of block ctr0 decide, the velocity comes from block ctr0 + natoms (kernels.jl:714-720);
// MODE 1 (random_velocities!): velocity from block ctr0 (kernels.jl:695-699).  A pending CM removal is applied on the way.
template <class T, int MODE>
__global__ void __launch_bounds__(256) k_redraw(int64_t n, typename Vec<T>::T4* vel, const int32_t* __restrict__ orig, StochP<T> P,
                                                const T* __restrict__ vcm, const double* __restrict__ cm_in, int n_cm_in) {
    T vc[3] = {T(0), T(0), T(0)};
    const bool sub = vcm != nullptr || cm_in != nullptr;
    if (cm_in) block_vcm<T>(cm_in, n_cm_in, vc);
    else if (vcm) { vc[0] = vcm[0]; vc[1] = vcm[1]; vc[2] = vcm[2]; }
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto v = vel[s];
        if (sub) { v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2]; }
        uint64_t ctr0 = (uint64_t)orig[s] + 1;
        bool draw = true;
        if constexpr (MODE == 0) {
            const U4 w = philox_u64(ctr0, P.ctr1, P.key);
            draw = ((((uint64_t)w.b) << 32) | w.a) < P.prob_u64;
            ctr0 += P.natoms;
        }
        if (draw) {
            T z[3];
            randn3<T>(ctr0, P.ctr1, P.key, P.natoms, z);
            const T sc = thermal_scale<T>(P.noise_kt, v.w);
            v.x = z[0] * sc; v.y = z[1] * sc; v.z = z[2] * sc;
        }
        if (draw || sub) vel[s] = v;
    }
}

__global__ void k_philox_probe(const uint32_t* __restrict__ in, uint32_t* out) {
    U4 c; c.a = in[0]; c.b = in[1]; c.c = in[2]; c.d = in[3];
    const U4 r = philox4x32_10(c, in[4], in[5]);
    out[0] = r.a; out[1] = r.b; out[2] = r.c; out[3] = r.d;
}

}  // namespace

template <class T>
void launch_langevin(hipStream_t s, int n_blocks, int64_t n, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel, const typename Vec<T>::T4* frc,
                     const int32_t* orig, const StochP<T>& P, const T* vcm, const double* cm_in, int n_cm_in, double* cm_out, const GridP<T>& G) {
    if (cm_out) hipLaunchKernelGGL((k_langevin<T, true>), dim3(n_blocks), dim3(256), 0, s, n, pos, vel, frc, orig, P, vcm, cm_in, n_cm_in, cm_out, G);
    else hipLaunchKernelGGL((k_langevin<T, false>), dim3(n_blocks), dim3(256), 0, s, n, pos, vel, frc, orig, P, vcm, cm_in, n_cm_in, cm_out, G);
}
template <class T>
void launch_redraw(hipStream_t s, int mode, int64_t n, typename Vec<T>::T4* vel, const int32_t* orig, const StochP<T>& P,
                   const T* vcm, const double* cm_in, int n_cm_in) {
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 1024);
    if (mode == 0) hipLaunchKernelGGL((k_redraw<T, 0>), dim3(nb), dim3(256), 0, s, n, vel, orig, P, vcm, cm_in, n_cm_in);
    else hipLaunchKernelGGL((k_redraw<T, 1>), dim3(nb), dim3(256), 0, s, n, vel, orig, P, vcm, cm_in, n_cm_in);
}
void philox_host(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint32_t* out4) {
    const U4 r = philox_u64(ctr0, ctr1, key);
    out4[0] = r.a; out4[1] = r.b; out4[2] = r.c; out4[3] = r.d;
}
void launch_philox_probe(hipStream_t s, const uint32_t* in, uint32_t* out) { hipLaunchKernelGGL(k_philox_probe, dim3(1), dim3(1), 0, s, in, out); }

template void launch_langevin<float>(hipStream_t, int, int64_t, float4*, float4*, const float4*, const int32_t*, const StochP<float>&, const float*, const double*, int, double*, const GridP<float>&);
template void launch_langevin<double>(hipStream_t, int, int64_t, double4*, double4*, const double4*, const int32_t*, const StochP<double>&, const double*, const double*, int, double*, const GridP<double>&);
template void launch_redraw<float>(hipStream_t, int, int64_t, float4*, const int32_t*, const StochP<float>&, const float*, const double*, int);
template void launch_redraw<double>(hipStream_t, int, int64_t, double4*, const int32_t*, const StochP<double>&, const double*, const double*, int);

}  // namespace mhip

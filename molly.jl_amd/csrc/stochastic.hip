// stochastic.hip — Philox4x32-10 noise, the fused Langevin update, Andersen / Maxwell-Boltzmann velocity draws (see stochastic.h)
#include "stochastic.h"

#include "kernels.h"
#include "philox.h"

namespace mhip {

namespace {

// One Langevin-middle step of the owned atoms, forces already at hand (simulators.jl:1171-1197):
//   v += (f/m)·dt ; x = muladd(dt/2, v, x) ; v = muladd(vel_scale, v, noise·noise_scale) ; x = muladd(dt/2, v, x) ; wrap
// plus the deferred remove_CM_motion! of the previous step in front and the Σ m v partials of this one behind.
template <class T, bool CM>
__global__ void __launch_bounds__(256) k_langevin(int64_t n, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel, const typename Vec<T>::T4* __restrict__ frc,
                                                  const int32_t* __restrict__ orig, StochP<T> P, const T* __restrict__ vcm, const double* __restrict__ cm_in,
                                                  int n_cm_in, double* cm_out, GridP<T> G, const typename Vec<T>::T4* __restrict__ frc_add) {
#pragma clang fp contract(off)
    T vc[3] = {T(0), T(0), T(0)};
    const bool sub = vcm != nullptr || cm_in != nullptr;
    if (cm_in) block_vcm<T>(cm_in, n_cm_in, vc);
    else if (vcm) { vc[0] = vcm[0]; vc[1] = vcm[1]; vc[2] = vcm[2]; }
    double px = 0, py = 0, pz = 0, m = 0;
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        auto v = vel[s]; auto p = pos[s]; auto f = frc[s];
        if (frc_add) { const auto g = frc_add[s]; f.x += g.x; f.y += g.y; f.z += g.z; }      // (the side array of a small system's step: bonded sums — what k_add_forces folded in a launch of its own)
        if (sub) { v.x -= vc[0]; v.y -= vc[1]; v.z -= vc[2]; }
        langevin_atom<T>(v, p, f, P, (uint64_t)orig[s] + 1, G);
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
                     const int32_t* orig, const StochP<T>& P, const T* vcm, const double* cm_in, int n_cm_in, double* cm_out, const GridP<T>& G, const typename Vec<T>::T4* frc_add) {
    if (cm_out) hipLaunchKernelGGL((k_langevin<T, true>), dim3(n_blocks), dim3(256), 0, s, n, pos, vel, frc, orig, P, vcm, cm_in, n_cm_in, cm_out, G, frc_add);
    else hipLaunchKernelGGL((k_langevin<T, false>), dim3(n_blocks), dim3(256), 0, s, n, pos, vel, frc, orig, P, vcm, cm_in, n_cm_in, cm_out, G, frc_add);
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

template void launch_langevin<float>(hipStream_t, int, int64_t, float4*, float4*, const float4*, const int32_t*, const StochP<float>&, const float*, const double*, int, double*, const GridP<float>&, const float4*);
template void launch_langevin<double>(hipStream_t, int, int64_t, double4*, double4*, const double4*, const int32_t*, const StochP<double>&, const double*, const double*, int, double*, const GridP<double>&, const double4*);
template void launch_redraw<float>(hipStream_t, int, int64_t, float4*, const int32_t*, const StochP<float>&, const float*, const double*, int);
template void launch_redraw<double>(hipStream_t, int, int64_t, double4*, const int32_t*, const StochP<double>&, const double*, const double*, int);

}  // namespace mhip

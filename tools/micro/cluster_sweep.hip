// cluster_sweep.hip — upper bound for a cluster-pair ("tile") Lennard-Jones kernel on gfx950, to set against k_forces.
//
// The layout of the reference's hot kernel (32 × 32 tiles, ext/MollyCUDAExt.jl:1595-2045) and of GROMACS' nbnxm, sized for wave64:
// an i-supercluster of 8 clusters × 8 atoms lives in LDS, one wave takes one j-cluster of 8 atoms at a time (lane = (j, i): 64 atom
// pairs per sweep), loops over the 8 i-clusters of the supercluster (mask bit per cluster pair), accumulates the i-forces of every
// i-cluster in registers (reduced over the j lanes once at the end), the j-forces over the whole i-loop (reduced over the i lanes
// once per j-cluster, then one atomic per j atom and component).  Newton's third law, no per-pair index, no gather: everything the
// per-atom list pays for.  Arithmetic as in k_forces' packed loop: two i-clusters side by side in the halves of 64-bit registers
// (v_pk_*_f32), one v_rcp_f32 per two pairs, clamped-fma cutoff.
//
// What is measured is the BEST case: every mask bit set (no skipped or half-empty sweeps), j-clusters streamed linearly, coordinates
// that keep all pairs inside the cutoff.  Pair evaluations per second of this loop × the evaluations a cluster list needs per atom
// (tools/cluster_fill.py: 220 for 8 × 8 clusters at the benchmark fluid's density, list radius 1.1 nm) = the fastest a cluster-pair
// force pass could be.    hipcc --offload-arch=gfx950 -O3 tools/micro/cluster_sweep.hip -o cluster_sweep && ./cluster_sweep
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

// sum over the 8 lanes of an aligned group with three DPP adds (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror): no LDS crossbar
__device__ inline float sum8(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    return v;
}

__global__ void __launch_bounds__(256) k_cluster(int n_super, int n_jc, const float4* __restrict__ xi, const float4* __restrict__ xj, const unsigned* __restrict__ /*masks*/,
                                                  float* fi_out, float* fj_out, float s2, float c24, float rc2) {
    __shared__ float4 l_xi[4][64];                 // per wave: the supercluster's 64 i-atoms
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 7, j = lane >> 3;   // the 8 lanes that share a j atom are neighbours: DPP reductions
    const int sc = blockIdx.x * 4 + wave;
    if (sc >= n_super) return;
    l_xi[wave][lane] = xi[(size_t)sc * 64 + lane];
    __builtin_amdgcn_s_waitcnt(0);
    const float rc2n = __int_as_float(__float_as_int(rc2) + 1);
    const v2f cut_a = {-0x1p100f, -0x1p100f}, cut_b = {rc2n * 0x1p100f, rc2n * 0x1p100f}, c48v = {c24 + c24, c24 + c24}, c24v = {c24, c24};
    v2f fix[4] = {}, fiy[4] = {}, fiz[4] = {};     // i-forces of the 8 i-clusters, two per register pair
    for (int jc = 0; jc < n_jc; ++jc) {
        const float4 pj = xj[((size_t)sc * n_jc + jc) * 8 + j];     // my j atom (8 distinct atoms per wave)
        // (a real kernel tests one mask bit per cluster pair here and skips or half-fills sweeps; the best case has every bit set, and
        // the branch is left out altogether: with it the compiler copies all 24 accumulators around every sweep)
        v2f fjx = {0.f, 0.f}, fjy = {0.f, 0.f}, fjz = {0.f, 0.f};
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {                             // i-clusters 2·c2 and 2·c2 + 1 side by side
            const float4 pa = l_xi[wave][(2 * c2) * 8 + i], pb = l_xi[wave][(2 * c2 + 1) * 8 + i];
            const v2f dx = (v2f){pj.x, pj.x} - (v2f){pa.x, pb.x}, dy = (v2f){pj.y, pj.y} - (v2f){pa.y, pb.y}, dz = (v2f){pj.z, pj.z} - (v2f){pa.z, pb.z};
            const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            const float t = __builtin_amdgcn_rcpf(r2.x * r2.y);
            const v2f inv = (v2f){r2.y, r2.x} * t;
            const v2f u = inv * s2, u3 = u * u * u;
            v2f f = __builtin_elementwise_fma(u3, c48v, -c24v) * u3 * inv;
            v2f in;
            asm("v_pk_fma_f32 %0, %1, %2, %3 clamp\n\ts_nop 0" : "=v"(in) : "v"(r2), "s"(cut_a), "v"(cut_b));
            f *= in;
            const v2f gx = dx * f, gy = dy * f, gz = dz * f;
            fix[c2] -= gx; fiy[c2] -= gy; fiz[c2] -= gz;             // force on i is −f·dr
            fjx += gx; fjy += gy; fjz += gz;                         // … and +f·dr on j
        }
        // j-forces: the two halves, then the 8 neighbouring lanes that share my j atom, one atomic per j atom and component
        const float ax = sum8(fjx.x + fjx.y), ay = sum8(fjy.x + fjy.y), az = sum8(fjz.x + fjz.y);
        if (i == 0) {
            float* dst = fj_out + (((size_t)sc * n_jc + jc) * 8 + j) * 3;
            atomicAdd(dst, ax); atomicAdd(dst + 1, ay); atomicAdd(dst + 2, az);
        }
    }
    // i-forces: sum over the 8 lanes that share my i atom (j = lane >> 3: xor 8, 16, 32), once per supercluster
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float ax = h ? fix[c2].y : fix[c2].x, ay = h ? fiy[c2].y : fiy[c2].x, az = h ? fiz[c2].y : fiz[c2].x;
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { ax += __shfl_xor(ax, o, 64); ay += __shfl_xor(ay, o, 64); az += __shfl_xor(az, o, 64); }
            if (j == 0) { float* dst = fi_out + ((size_t)sc * 64 + (2 * c2 + h) * 8 + i) * 3; dst[0] = ax; dst[1] = ay; dst[2] = az; }
        }
}

int main() {
    const int n_super = 16384, n_jc = 96;           // 1 M i-atoms, 96 j-clusters each: 8 × 64 × 96 = 49 152 pair evaluations per wave
    std::vector<float4> hxi((size_t)n_super * 64), hxj((size_t)n_super * n_jc * 8);
    std::vector<unsigned> hm((size_t)n_super * n_jc, 0xffu);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto& p : hxi) p = make_float4(rnd() * 0.5f, rnd() * 0.5f, rnd() * 0.5f, 0.f);
    for (auto& p : hxj) p = make_float4(0.9f + rnd() * 0.3f, rnd() * 0.3f, rnd() * 0.3f, 0.f);      // 0.45 … 1.3 nm away: inside and outside the cutoff
    float4 *xi, *xj; unsigned* m; float *fi, *fj;
    hipMalloc(&xi, hxi.size() * sizeof(float4)); hipMalloc(&xj, hxj.size() * sizeof(float4)); hipMalloc(&m, hm.size() * 4);
    hipMalloc(&fi, (size_t)n_super * 64 * 3 * 4); hipMalloc(&fj, (size_t)n_super * n_jc * 8 * 3 * 4);
    hipMemcpy(xi, hxi.data(), hxi.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMemcpy(xj, hxj.data(), hxj.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    hipMemset(fj, 0, (size_t)n_super * n_jc * 8 * 3 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_cluster, dim3(n_super / 4), dim3(256), 0, 0, n_super, n_jc, xi, xj, m, fi, fj, 0.34f * 0.34f, 24.f * 0.997f, 1.0f);
    hipEventRecord(a);
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k_cluster, dim3(n_super / 4), dim3(256), 0, 0, n_super, n_jc, xi, xj, m, fi, fj, 0.34f * 0.34f, 24.f * 0.997f, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double evals = (double)n_super * n_jc * 8 * 64;
    std::vector<float> hf(12);
    hipMemcpy(hf.data(), fi, 12 * 4, hipMemcpyDeviceToHost);
    printf("cluster sweep (8x8, all masks set): %.3f ms for %.1f M pair evaluations = %.2f T evaluations/s   (check %g %g %g)\n", ms, evals / 1e6, evals / (ms * 1e-3) / 1e12, hf[0], hf[1], hf[2]);
    printf("=> a cluster-pair force pass of the 1M-atom fluid (220 evaluations per atom, tools/cluster_fill.py) would take at least %.1f us;\n", 220e6 / (evals / (ms * 1e-3)) * 1e6);
    printf("   with 4x4 clusters (151 per atom) at the same rate %.1f us — half-empty sweeps, j-cluster gathers and mask tests not counted.\n", 151e6 / (evals / (ms * 1e-3)) * 1e6);
    return 0;
}

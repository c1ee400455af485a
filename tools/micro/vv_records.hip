// vv_records.hip — is the integrator launch (k_vv_mid: second kick, first kick, drift, wrap) worth 12-byte velocity / force records?
// (VERDICT rounds 3 and 4, item 5.)  The same arithmetic over three layouts, 512 workgroups of 256 lanes with the engine's one-atom-ahead fetch:
//   A  pos T4 r/w | vel T4 {v, m} r/w | frc T4 r                      80 B per atom  (the engine's layout)
//   B  pos T4 r/w | vel 3 × f32 r/w + mass f32 r | frc 3 × f32 r      72 B per atom
//   C  as B, one mass for every atom (kernel argument)                 68 B per atom
// N = 1 000 000 and 262 144.  Build: hipcc --offload-arch=gfx950 -O3 vv_records.hip -o vv_records
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline float wrapf(float x, float L) { return x - floorf(x / L) * L; }

__global__ void __launch_bounds__(256) k_a(int n, float4* pos, float4* vel, const float4* __restrict__ frc, float dt, float dt2, float L) {
    const int stride = gridDim.x * blockDim.x;
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    float4 v, f, p;
    if (s < n) { v = vel[s]; f = frc[s]; p = pos[s]; }
    for (; s < n; s += stride) {
        const float4 v0 = v, f0 = f, p0 = p;
        if (s + stride < n) { v = vel[s + stride]; f = frc[s + stride]; p = pos[s + stride]; }
        float4 vv = v0, pp = p0;
        const float im = 1.f / v0.w, kx = f0.x * im * dt2, ky = f0.y * im * dt2, kz = f0.z * im * dt2;
        vv.x += kx; vv.y += ky; vv.z += kz; vv.x += kx; vv.y += ky; vv.z += kz;
        pp.x = wrapf(pp.x + vv.x * dt, L); pp.y = wrapf(pp.y + vv.y * dt, L); pp.z = wrapf(pp.z + vv.z * dt, L);
        pos[s] = pp; vel[s] = vv;
    }
}
template <bool UNIFORM>
__global__ void __launch_bounds__(256) k_b(int n, float4* pos, float* vel3, const float* __restrict__ mass, const float* __restrict__ frc3, float dt, float dt2, float L, float m_all) {
    const int stride = gridDim.x * blockDim.x;
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    float vx, vy, vz, fx, fy, fz, m = m_all; float4 p;
    auto fetch = [&](int a) { vx = vel3[3 * (size_t)a]; vy = vel3[3 * (size_t)a + 1]; vz = vel3[3 * (size_t)a + 2]; fx = frc3[3 * (size_t)a]; fy = frc3[3 * (size_t)a + 1]; fz = frc3[3 * (size_t)a + 2];
                              if (!UNIFORM) m = mass[a]; p = pos[a]; };
    if (s < n) fetch(s);
    for (; s < n; s += stride) {
        const float ax = vx, ay = vy, az = vz, bx = fx, by = fy, bz = fz, m0 = m; const float4 p0 = p;
        if (s + stride < n) fetch(s + stride);
        const float im = 1.f / m0, kx = bx * im * dt2, ky = by * im * dt2, kz = bz * im * dt2;
        float ux = ax + kx + kx, uy = ay + ky + ky, uz = az + kz + kz;
        float4 pp = p0;
        pp.x = wrapf(pp.x + ux * dt, L); pp.y = wrapf(pp.y + uy * dt, L); pp.z = wrapf(pp.z + uz * dt, L);
        pos[s] = pp; vel3[3 * (size_t)s] = ux; vel3[3 * (size_t)s + 1] = uy; vel3[3 * (size_t)s + 2] = uz;
    }
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int n : {1000000, 262144}) {
        float4 *pos, *vel, *frc; float *vel3, *frc3, *mass;
        CK(hipMalloc(&pos, (size_t)n * 16)); CK(hipMalloc(&vel, (size_t)n * 16)); CK(hipMalloc(&frc, (size_t)n * 16));
        CK(hipMalloc(&vel3, (size_t)n * 12)); CK(hipMalloc(&frc3, (size_t)n * 12)); CK(hipMalloc(&mass, (size_t)n * 4));
        CK(hipMemset(pos, 0, (size_t)n * 16)); CK(hipMemset(vel, 0x3f, (size_t)n * 16)); CK(hipMemset(frc, 0, (size_t)n * 16));
        CK(hipMemset(vel3, 0, (size_t)n * 12)); CK(hipMemset(frc3, 0, (size_t)n * 12)); CK(hipMemset(mass, 0x3f, (size_t)n * 4));
        for (int blocks : {256, 512, 1024}) {
            const int nb = blocks < (n + 255) / 256 ? blocks : (n + 255) / 256, reps = 400;
            float ms[3];
            for (int v = 0; v < 3; ++v) for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) {
                    if (v == 0) hipLaunchKernelGGL(k_a, dim3(nb), dim3(256), 0, 0, n, pos, vel, frc, 0.002f, 0.001f, 36.f);
                    else if (v == 1) hipLaunchKernelGGL(k_b<false>, dim3(nb), dim3(256), 0, 0, n, pos, vel3, mass, frc3, 0.002f, 0.001f, 36.f, 39.9f);
                    else hipLaunchKernelGGL(k_b<true>, dim3(nb), dim3(256), 0, 0, n, pos, vel3, mass, frc3, 0.002f, 0.001f, 36.f, 39.9f);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[v], e0, e1));
            }
            std::printf("n %8d blocks %5d | A 80 B/atom %7.2f us (%5.2f TB/s) | B 72 B/atom %7.2f us (%5.2f TB/s) | C 68 B/atom %7.2f us (%5.2f TB/s)\n", n, nb,
                        ms[0] * 1e3 / reps, 80.0 * n / (ms[0] * 1e-3 / reps) * 1e-12, ms[1] * 1e3 / reps, 72.0 * n / (ms[1] * 1e-3 / reps) * 1e-12, ms[2] * 1e3 / reps, 68.0 * n / (ms[2] * 1e-3 / reps) * 1e-12);
        }
        (void)hipFree(pos); (void)hipFree(vel); (void)hipFree(frc); (void)hipFree(vel3); (void)hipFree(frc3); (void)hipFree(mass);
    }
    return 0;
}

// scratch_cost.hip — what does a kernel pay for OWNING scratch (a private segment in its descriptor), whether or not a lane ever touches it?
// (round 5: the packed pair pass with 44-60 bytes of spills OUTSIDE its loops ran 16-25 % slower, profiles/r05_force_ab.txt §10; the fused first launch of the
// 6mrr step, k_pair_spread_bonded, owns 68 bytes that no instruction uses.)  The same arithmetic in two kernels: k_plain, and k_owner with a 17-word private
// array that is written only under a condition that never holds.  Grids of many short workgroups (the pair pass's shape: 512-lane blocks, ≈ 20 µs each) and of
// one resident round.  Build: hipcc --offload-arch=gfx950 -O3 scratch_cost.hip -o scratch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <bool OWNER>
__global__ void __launch_bounds__(512) k_work(float* buf, int iters, int never) {
    float acc = (float)threadIdx.x;
    if constexpr (OWNER) {
        volatile float priv[17];
        if (never) { for (int i = 0; i < 17; ++i) priv[i] = acc + i; acc += priv[never % 17]; }
    }
    for (int i = 0; i < iters; ++i) acc = acc * 1.0001f + 0.5f;
    if (acc == 123.456f) buf[blockIdx.x] = acc;
}
int main() {
    float* buf; CK(hipMalloc(&buf, 1 << 20));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {1024, 4096, 16384}) for (int iters : {2000, 20000}) {
        float ms[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) for (int owner = 0; owner < 2; ++owner) {
            auto go = [&] { if (owner) hipLaunchKernelGGL(k_work<true>, dim3(blocks), dim3(512), 0, s, buf, iters, 0); else hipLaunchKernelGGL(k_work<false>, dim3(blocks), dim3(512), 0, s, buf, iters, 0); };
            for (int w = 0; w < 5; ++w) go();
            CK(hipEventRecord(e0, s)); for (int q = 0; q < 50; ++q) go(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[owner] = t / 50;
        }
        std::printf("%6d blocks x 512 lanes, %5d dependent fma: plain %8.2f us   owning 68 bytes of scratch %8.2f us   (%+.1f %%)\n", blocks, iters, ms[0] * 1e3, ms[1] * 1e3, 100.0 * (ms[1] / ms[0] - 1));
    }
    return 0;
}

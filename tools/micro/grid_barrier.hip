// grid_barrier.hip — what does a barrier across a SMALL grid cost on gfx950?
// G workgroups of 256 lanes (G <= 64, all resident), per round: every workgroup writes `bytes` of fresh data, arrives at a counter with an
// agent-scope release, spins with agent-scope acquire loads until all G have arrived, then reads the slice its RIGHT neighbour wrote (so the
// barrier really orders data across compute units and XCDs).  Reports µs per round for several G and slice sizes, and the same loop without
// the barrier (the data movement alone).  Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool BARRIER>
__global__ void __launch_bounds__(256) k_rounds(float* buf, int words_per_wg, int rounds, unsigned int* ctr, float* sink, unsigned long long* bad) {
    const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * G + b) * words_per_wg;
        for (int w = t; w < words_per_wg; w += 256) mine[w] = (float)(r * 1000 + b) + acc * 1e-30f;
        if (BARRIER) {
            __threadfence();
            __syncthreads();
            if (t == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int want = (unsigned int)(r + 1) * (unsigned int)G;
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
        }
        const int nb = (b + 1) % G;
        const float* theirs = buf + ((size_t)(r & 1) * G + nb) * words_per_wg;
        float s = 0.f;
        for (int w = t; w < words_per_wg; w += 256) s += __builtin_nontemporal_load(theirs + w);
        if (BARRIER && t == 0 && theirs[0] != (float)(r * 1000 + nb)) atomicAdd(bad, 1ull);
        acc += s;
    }
    if (acc == 12345.678f) sink[b] = acc;
}

int main() {
    const int rounds = 2000;
    float* buf; unsigned int* ctr; float* sink; unsigned long long* bad;
    CK(hipMalloc(&buf, (size_t)2 * 64 * 65536 * sizeof(float))); CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&sink, 64 * 4)); CK(hipMalloc(&bad, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::printf("# workgroups  bytes/wg  us/round(with barrier)  us/round(no barrier)  barrier alone  stale reads\n");
    for (int G : {8, 16, 32, 46, 64}) for (int words : {64, 1024, 4096, 16384}) {
        float ms[2];
        unsigned long long hbad = 0;
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(ctr, 0, 4)); CK(hipMemset(bad, 0, 8));
            for (int rep = 0; rep < 2; ++rep) {      // (first repetition warms up)
                CK(hipMemset(ctr, 0, 4));
                CK(hipEventRecord(e0));
                if (v == 0) hipLaunchKernelGGL(k_rounds<true>, dim3(G), dim3(256), 0, 0, buf, words, rounds, ctr, sink, bad);
                else hipLaunchKernelGGL(k_rounds<false>, dim3(G), dim3(256), 0, 0, buf, words, rounds, ctr, sink, bad);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[v], e0, e1));
            }
            if (v == 0) CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
        }
        std::printf("%4d %8d %10.3f %10.3f %10.3f %llu\n", G, words * 4, ms[0] * 1e3 / rounds, ms[1] * 1e3 / rounds, (ms[0] - ms[1]) * 1e3 / rounds, hbad);
    }
    return 0;
}

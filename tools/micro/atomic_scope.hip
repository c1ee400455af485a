// Global float atomics onto a PME-sized mesh (108 k points) from 250 workgroups: agent scope on one shared mesh (what the spread does)
// against workgroup scope on a mesh private to the XCD the workgroup runs on (the atomic then completes in that XCD's L2).
// The private copies are summed afterwards; the total must be exact (every add is 1.0f).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int MESH = 46 * 46 * 51;
__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }   // HW_REG_XCC_ID, bits 3:0
template <int MODE> __global__ void k(float* mesh, int per_thread, int* seen) {
    float* m = mesh;
    if (MODE == 1) { const int x = xcc_id(); m = mesh + (size_t)x * MESH; if (threadIdx.x == 0) atomicOr(&seen[x], 1); }
    // a batch = a compact box of the mesh, like the spread's sub-mesh flush
    const int base = (blockIdx.x * 431) % (MESH - 4096);
    for (int i = 0; i < per_thread; ++i) {
        const int a = base + ((threadIdx.x + 256 * i) % 4096);
        if (MODE == 0) atomicAdd(m + a, 1.0f);
        else __hip_atomic_fetch_add(m + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
int main() {
    float* d; int* seen; hipMalloc(&d, sizeof(float) * MESH * 8); hipMalloc(&seen, 32);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int per = 8, nblk = 250;   // 250 x 256 x 8 = 512 k atomics
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f; double total = 0;
        for (int rep = 0; rep < 6; ++rep) {
            hipMemset(d, 0, sizeof(float) * MESH * 8); hipMemset(seen, 0, 32); hipDeviceSynchronize();
            hipEventRecord(a);
            if (mode == 0) k<0><<<nblk, 256>>>(d, per, seen); else k<1><<<nblk, 256>>>(d, per, seen);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            std::vector<float> h((size_t)MESH * 8); hipMemcpy(h.data(), d, sizeof(float) * MESH * 8, hipMemcpyDeviceToHost);
            total = 0; for (float v : h) total += v;
        }
        int hs[8]; hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost);
        int nx = 0; for (int i = 0; i < 8; ++i) nx += hs[i];
        printf("%s: %.1f us for %d atomics, sum %.0f (expected %d), XCDs seen %d\n", mode == 0 ? "agent scope, one mesh" : "workgroup scope, mesh per XCD", best * 1e3, nblk * 256 * per, total, nblk * 256 * per, nx);
    }
    return 0;
}

// Do two independent kernels on ONE stream overlap when the second is launched with hipExtAnyOrderLaunch (no AQL barrier bit)?
// hipcc --offload-arch=gfx950 -O3 tools/micro/anyorder.hip -o anyorder && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(float* out, int iters) {
    float a = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) a = a * 1.000001f + 1e-7f;
    if (a == 123.f) out[0] = a;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 200000;
    auto run = [&](int flags, const char* name) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, s);
            for (int k = 0; k < 10; ++k) {
                hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, 0, d, iters);
                hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, flags, d, iters);
            }
            hipEventRecord(b, s); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%-28s 20 kernels of 64 blocks: %.3f ms\n", name, ms);
        }
    };
    run(0, "in order");
    run(hipExtAnyOrderLaunch, "second of each pair any-order");
    return 0;
}

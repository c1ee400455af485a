// graph_launch.hip — does a hipGraph of a small step's dependent kernels run with fewer gaps than the same launches issued one by one?
// (the 6mrr_pme step is eight dependent dispatches of 5-26 µs; its rocprofv3 timeline shows 72 µs of kernels in a 78 µs step.)
// A "step" here = 8 dependent kernels with ≈ 400-byte argument blocks that each keep 64 workgroups busy for `busy` µs.  Timed: N steps as plain launches,
// N launches of a graph holding one step, N/4 launches of a graph holding four steps.  Build: hipcc --offload-arch=gfx950 -O3 graph_launch.hip -o graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Fat { float a[96]; };      // 384 bytes of by-value arguments, like GridP + InterP + pointers
__global__ void __launch_bounds__(256) k_busy(float* buf, Fat f, int ticks) {
    const unsigned long long t0 = wall_clock64();
    float acc = f.a[threadIdx.x % 96];
    while (wall_clock64() - t0 < (unsigned long long)ticks) acc = acc * 1.0001f + 0.5f;
    if (acc == 123.456f) buf[blockIdx.x] = acc;
}
int main() {
    float* buf; CK(hipMalloc(&buf, 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Fat f{}; for (int i = 0; i < 96; ++i) f.a[i] = (float)i;
    const int N = 2000;
    for (int busy_us : {2, 5, 8}) {
        const int ticks = busy_us * 100;
        auto step = [&](hipStream_t st) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, st, buf, f, ticks); };
        auto wall = [&](auto&& body) { CK(hipStreamSynchronize(s)); auto t0 = std::chrono::steady_clock::now(); body(); auto t1 = std::chrono::steady_clock::now(); (void)hipStreamSynchronize(s);
                                       auto t2 = std::chrono::steady_clock::now(); std::printf("   host enqueue %7.2f us/step, total %7.2f us/step", std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N); return 0; };
        std::printf("kernels of %d us, 8 per step (floor %d us/step):\n  plain launches:", busy_us, 8 * busy_us);
        wall([&] { for (int i = 0; i < N; ++i) step(s); }); std::printf("\n");
        for (int per : {1, 4}) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int q = 0; q < per; ++q) step(s);
            CK(hipStreamEndCapture(s, &g));
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const double inst = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            std::printf("  graph of %d step(s) (instantiate %.0f us):", per, inst);
            wall([&] { for (int i = 0; i < N / per; ++i) (void)hipGraphLaunch(ge, s); }); std::printf("\n");
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    }
    return 0;
}

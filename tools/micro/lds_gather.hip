// Micro-benchmark: random gathers of 3 floats from an LDS tile, four layouts/instructions (timing experiment for k_forces).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void k(const unsigned* __restrict__ idx, int rows, int tile_n, float* out) {
    extern __shared__ float lds[];
    const int stride = (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 6) ? 4 : (MODE == 5 ? 4 : 3);
    for (int t = threadIdx.x; t < tile_n * stride; t += blockDim.x) lds[t] = (float)(t % 97) * 0.01f;
    __syncthreads();
    float sx = 0, sy = 0, sz = 0;
    const unsigned* my = idx + (size_t)blockIdx.x * rows * blockDim.x + threadIdx.x;
    for (int r = 0; r < rows; ++r) {
        unsigned s = my[(size_t)r * blockDim.x];
        float x, y, z;
        if (MODE == 0) { const float4 v = reinterpret_cast<const float4*>(lds)[s]; x = v.x; y = v.y; z = v.z; }
        else if (MODE == 1) { v3f v; asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(4 * 3 * s)) : "memory"); x = v.x; y = v.y; z = v.z; }
        else if (MODE == 2) { const float* p = lds + 3 * s; x = p[0]; y = p[1]; z = p[2]; }
        else if (MODE == 3) { v4f v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(4 * 4 * s)) : "memory"); x = v.x; y = v.y; z = v.z; }
        if (MODE == 4) { const float4 v = reinterpret_cast<const float4*>(lds)[s]; x = v.x; y = v.y; z = v.z + v.w; }
        if (MODE == 5) { const float* p = lds + 3 * s; x = p[0]; y = p[1]; z = p[2] + lds[3 * tile_n + 3 + s]; }
        if (MODE == 6) { const float2* p = reinterpret_cast<const float2*>(lds) + 2 * s; const float2 a = p[0], b = p[1]; x = a.x; y = a.y; z = b.x + b.y; }
        sx += x; sy += y; sz += z;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sx + sy + sz;
}
int main() {
    const int blocks = 2048, threads = 512, rows = 600, tile_n = 3600;
    std::vector<unsigned> h((size_t)blocks * rows * threads);
    srand(1);
    // neighbour-like indices: lane l of a block walks slots around (l*7 + r*6) with jitter, as sorted lists do
    for (int b = 0; b < blocks; ++b) for (int r = 0; r < rows; ++r) for (int t = 0; t < threads; ++t)
        h[((size_t)b * rows + r) * threads + t] = (unsigned)(((t % 256) * 5 + r * 5 + rand() % 900) % tile_n);
    unsigned* d; float* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, (size_t)blocks * threads * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
    auto run = [&](auto kern, const char* name) {
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), tile_n * 16, 0, d, rows, tile_n, o);
        hipEventRecord(a);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), tile_n * 16, 0, d, rows, tile_n, o);
        hipEventRecord(b2); hipEventSynchronize(b2);
        float ms; hipEventElapsedTime(&ms, a, b2);
        printf("%-28s %.3f ms per launch (%.1f G gathers/s)\n", name, ms / 5, (double)blocks * threads * rows / (ms / 5 * 1e-3) / 1e9);
    };
    run(k<0>, "stride16 ds_read_b96");
    run(k<1>, "stride12 ds_read_b96 (asm)");
    run(k<2>, "stride12 read2_b32+b32");
    run(k<3>, "stride16 ds_read_b128");
    run(k<4>, "float4 all used (compiler)");
    run(k<5>, "xyz packed + q separate");
    run(k<6>, "float4 as 2 x float2");
    return 0;
}

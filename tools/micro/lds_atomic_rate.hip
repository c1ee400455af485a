// Rate of LDS atomic adds on gfx950: ds_add_f32 against ds_add_u32 / ds_add_u64, in the access pattern of the PME charge spreading
// (a half-wave per atom: 25 lanes on a 5 x 5 patch of an ey x ez plane, five adds along x each).
#include <hip/hip_runtime.h>
#include <cstdio>
template <class V> __global__ void k(V* out, int rounds, long long* cyc) {
    __shared__ V box[6144];
    for (int c = threadIdx.x; c < 6144; c += 256) box[c] = V(0);
    __syncthreads();
    const int sub = threadIdx.x & 31, hw = threadIdx.x >> 5, ey = 12, ez = 12;
    const long long t0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        const int bx = (hw * 3 + r) % 7, by = (hw * 5 + r * 2) % 7, bz = (hw + r * 3) % 7;
        if (sub < 25) {
            const int iy = sub / 5, iz = sub - iy * 5;
            V* col = box + (by + iy) * ez + (bz + iz);
#pragma unroll
            for (int ix = 0; ix < 5; ++ix) atomicAdd(col + (bx + ix) * ey * ez, V(1));
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    V s = 0; for (int c = threadIdx.x; c < 6144; c += 256) s += box[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class V> void run(const char* name) {
    V* d; long long* c; hipMalloc(&d, sizeof(V) * 256 * 256); hipMalloc(&c, 8 * 256);
    const int rounds = 8;     // 8 atoms per half-wave = a 64-atom batch
    k<V><<<250, 256>>>(d, rounds, c); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k<V><<<250, 256>>>(d, rounds, c); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[256]; hipMemcpy(h, c, 8 * 250, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 250; ++i) m += h[i];
    printf("%-12s kernel %.1f us, add phase %.0f clock64 ticks per block (100 MHz: %.2f us) for %d adds per block\n", name, ms * 1e3, m / 250, m / 250 / 100.0, 8 * 25 * 5 * rounds);
}
int main() { run<float>("ds_add_f32"); run<unsigned>("ds_add_u32"); run<unsigned long long>("ds_add_u64"); run<int>("ds_add_i32"); return 0; }

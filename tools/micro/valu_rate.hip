// Micro-benchmark: issue cost of the VALU instruction kinds the packed pair loop is made of, at 8 / 4 waves per SIMD (timing experiment
// for k_forces: is a v_pk_*_f32 one issue slot or two, what do v_rcp_f32 and the SDWA address adds cost).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(512) k(float* out, int iters, float seed) {
    v2f a[8]; float s[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = (v2f){seed + i + threadIdx.x, seed * 0.5f + i}; s[i] = seed + i * 0.25f + threadIdx.x; u[i] = threadIdx.x * 7u + i; }
    const v2f m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f}; const unsigned seed_u = (unsigned)seed * 977u + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m.x), "v"(c.x));
            if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
            if (MODE == 3) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 5) asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(u[i]));
            if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(m.x));
            if (MODE == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(0x05040100u));
            if (MODE == 9) { unsigned long long w = ((unsigned long long)u[i] << 32) | u[(i + 1) & 7]; asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(w) : "v"(u[(i + 2) & 7] & 31u)); u[i] = (unsigned)(w >> 32); }
            if (MODE == 10) asm volatile("v_rsq_f32 %0, %0" : "+v"(s[i]));
            if (MODE == 11) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
            if (MODE == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (MODE == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (MODE == 14) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (MODE == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(0x5555555555555555ull));
            if (MODE == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(seed_u) : );       // second operand loop-invariant: independent chains
            if (MODE == 17) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(seed_u) : "vcc");
            if (MODE == 18) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(seed_u));
        }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float* o; hipMalloc(&o, 4096 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](auto kern, const char* name, int blocks_per_cu) {
        const int blocks = 256 * blocks_per_cu;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, o, 10, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, o, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD: blocks_per_cu * 8 waves / 4 SIMDs * iters * 8
        const double winst = (double)blocks_per_cu * 2 * iters * 8;
        printf("%-26s %d blocks/CU (%2d waves/SIMD): %.3f ms  => %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks_per_cu, blocks_per_cu * 2, ms, ms * 1e-3 * 2.4e9 / winst);
    };
    for (int bpc : {4, 2, 1}) {
        run(k<0>, "v_pk_fma_f32", bpc); run(k<4>, "v_pk_mul_f32", bpc); run(k<7>, "v_pk_add_f32", bpc); run(k<1>, "v_fma_f32", bpc); run(k<6>, "v_mul_f32", bpc);
        run(k<2>, "v_rcp_f32", bpc); run(k<3>, "v_add_u32_sdwa", bpc); run(k<5>, "v_and_b32", bpc);
        if (bpc == 4) {
            run(k<8>, "v_perm_b32", bpc); run(k<9>, "v_lshlrev_b64 (+ pack)", bpc); run(k<10>, "v_rsq_f32", bpc); run(k<11>, "v_exp_f32", bpc);
            run(k<12>, "v_alignbit_b32", bpc); run(k<13>, "v_cndmask_b32 (vcc)", bpc); run(k<14>, "v_bcnt_u32_b32", bpc);
            run(k<15>, "v_cndmask_b32_e64 (sgpr)", bpc); run(k<16>, "v_cndmask_b32 indep.", bpc); run(k<17>, "v_cmp + v_cndmask (2 instr)", bpc); run(k<18>, "v_max_u32", bpc);
        }
    }
    return 0;
}

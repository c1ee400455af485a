// v_perm_b32 byte numbering as __builtin_amdgcn_perm(a, b, sel) sees it: which operand do selector values 0-3 address?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    out[0] = __builtin_amdgcn_perm(0x77665544u, 0x33221100u, 0x07060100u);
    out[1] = __builtin_amdgcn_perm(0x77665544u, 0x33221100u, 0x0c0c0504u);
    unsigned long long p = 0x0000000000001234ull; int sh = 48;
    out[2] = (unsigned)((p << sh) >> 32);
}
int main() {
    unsigned* d; hipMalloc(&d, 16); k<<<1, 1>>>(d); unsigned h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%08x %08x %08x\n", h[0], h[1], h[2]);
    return 0;
}

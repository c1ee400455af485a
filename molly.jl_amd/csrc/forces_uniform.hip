// forces_uniform.hip — the fp32 one-type Lennard-Jones instantiations of k_forces (uniform σ, ϵ; no Coulomb; forces only), in their
// own translation unit because they want the SLP vectoriser OFF: their inner loop is hand-written on float2 values (kernels.h,
// walk_rows) and auto-vectorisation on top of that re-pairs the operands and pays for it in v_mov shuffles and s_nops.
#include "kernels.h"

namespace mhip {

void launch_forces_uniform_f32(const ForceArgs<float>& A, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream) {
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(A.blocks_per_xcd * 8), dim3(threads), lds, stream, A);
    };
    if (prune) { if (seg) go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, true, true>); else go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, false, true>); }
    else { if (seg) go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, true, false>); else go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, false, false>); }
}

}  // namespace mhip

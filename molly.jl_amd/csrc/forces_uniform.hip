// forces_uniform.hip — the fp32 one-type Lennard-Jones instantiations of k_forces (uniform σ, ϵ; no Coulomb; forces only), in their
// own translation unit because they want the SLP vectoriser OFF: their inner loop is hand-written on float2 values (kernels.h,
// walk_rows) and auto-vectorisation on top of that re-pairs the operands and pays for it in v_mov shuffles and s_nops.
#include "kernels.h"

namespace mhip {

template <int STRIDE>
static void launch_stride(const ForceArgs<float>& A, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream, bool step, bool halo, bool lang) {
    auto go = [&](auto kern, unsigned extra) {
        if (lds > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(A.blocks_per_xcd * 8 + extra), dim3(threads), lds, stream, A);
    };
    if (step && halo) { go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, false, false, STRIDE, true, true>, 1u); return; }      // the fused step of a ghosted sub-domain (kernels.h, HaloStep)
    if (step && lang) { go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, false, false, STRIDE, true, false, true>, 1u); return; }      // … with the Langevin-middle update in the epilogue (mhip_langevin_run)
    if (step) { go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, false, false, STRIDE, true>, 1u); return; }      // (plain, unsegmented passes only: the engine asks for nothing else)
    if (prune) { if (seg) go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, true, true, STRIDE>, 0u); else go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, false, true, STRIDE>, 0u); }
    else { if (seg) go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, true, false, STRIDE>, 0u); else go(k_forces<float, LJ_DIST_UNIFORM, MHIP_COUL_NONE, false, false, false, false, STRIDE>, 0u); }
}

// A.soa = the stride of the x[] / y[] / z[] tile arrays the caller sized the LDS for (one of SOA_STRIDES), or 0: generic float4 tile
// step: the STEP variant (the integrator in the epilogue; A.vel … A.snap_b filled in), packed plain passes only (A.soa != 0, no segments, no prune)
// halo: … of a ghosted sub-domain (A.H filled in): waits for the peers, stages ghosts from the receive half, sends in the epilogue
void launch_forces_uniform_f32(const ForceArgs<float>& A, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream, bool step, bool halo, bool lang) {
    if (step && (seg || prune || !A.soa)) throw ApiError{MHIP_ERR_INVALID, "the fused step exists for the packed plain pass only"};
    if (lang && (!step || halo)) throw ApiError{MHIP_ERR_INVALID, "the Langevin update rides in the fused single-domain step only"};
    if (A.soa == SOA_STRIDES[0]) launch_stride<SOA_STRIDES[0]>(A, seg, prune, lds, threads, stream, step, halo, lang);
    else if (A.soa == SOA_STRIDES[1]) launch_stride<SOA_STRIDES[1]>(A, seg, prune, lds, threads, stream, step, halo, lang);
    else launch_stride<SOA_STRIDES[2]>(A, seg, prune, lds, threads, stream, step, halo, lang);      // also A.soa == 0: the instantiation's generic loop
}

}  // namespace mhip

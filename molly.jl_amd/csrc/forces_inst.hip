// forces_inst.hip — the k_forces instantiations of ONE (precision, Coulomb kind), chosen with -DMHIP_INST_T=float|double and
// -DMHIP_INST_COUL=0..4 (4 = Ewald with the libm erfc): ten small translation units instead of one that takes minutes (the pair kernel has 48 variants per
// precision and Coulomb kind: LJ mode × {forces, forces + prune, energy} × minimum-image mode × segmented tile).  The fp32 one-type
// LJ variants live in forces_uniform.hip (SLP vectoriser off).
#include "kernels.h"
#include "forces_launch.h"

#ifndef MHIP_INST_T
#error "compile with -DMHIP_INST_T=float|double -DMHIP_INST_COUL=0|1|2|3|4"
#endif

namespace mhip {

namespace {
using T = MHIP_INST_T;
constexpr int COULM = MHIP_INST_COUL;

template <int LJM, bool ENERGY, bool MINIMG, bool SEG, bool PRUNE>
void launch_one(const ForceArgs<T>& A, size_t lds, unsigned threads, hipStream_t stream) {
    if constexpr (std::is_same<T, float>::value && LJM == LJ_DIST_UNIFORM && COULM == MHIP_COUL_NONE && !ENERGY && !MINIMG) {
        launch_forces_uniform_f32(reinterpret_cast<const ForceArgs<float>&>(A), SEG, PRUNE, lds, threads, stream);   // (T is float here)
    } else {
        auto kern = k_forces<T, LJM, COULM, ENERGY, MINIMG, SEG, PRUNE>;
        if (lds > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(A.blocks_per_xcd * 8), dim3(threads), lds, stream, A);
    }
}
template <int LJM, bool ENERGY, bool PRUNE>
void launch_geom(const ForceArgs<T>& A, bool minimg, bool seg, size_t lds, unsigned threads, hipStream_t stream) {
    if (minimg) { if (seg) launch_one<LJM, ENERGY, true, true, PRUNE>(A, lds, threads, stream); else launch_one<LJM, ENERGY, true, false, PRUNE>(A, lds, threads, stream); }
    else { if (seg) launch_one<LJM, ENERGY, false, true, PRUNE>(A, lds, threads, stream); else launch_one<LJM, ENERGY, false, false, PRUNE>(A, lds, threads, stream); }
}
template <int LJM>
void launch_mode(const ForceArgs<T>& A, bool energy, bool minimg, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream) {
    if (energy) launch_geom<LJM, true, false>(A, minimg, seg, lds, threads, stream);       // an energy pass never prunes
    else if (prune) launch_geom<LJM, false, true>(A, minimg, seg, lds, threads, stream);
    else launch_geom<LJM, false, false>(A, minimg, seg, lds, threads, stream);
}
}  // namespace

template <>
void launch_forces_tc<T, COULM>(const ForceArgs<T>& A, int ljm, bool energy, bool minimg, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream) {
    if (threads > (unsigned)BlockLimits<T>::max_threads) throw ApiError{MHIP_ERR_INVALID, "pair kernel launched with more lanes than its launch bound"};
    switch (ljm) {
    case LJ_OFF: launch_mode<LJ_OFF>(A, energy, minimg, seg, prune, lds, threads, stream); break;
    case LJ_DIST: launch_mode<LJ_DIST>(A, energy, minimg, seg, prune, lds, threads, stream); break;
    case LJ_DIST_UNIFORM: launch_mode<LJ_DIST_UNIFORM>(A, energy, minimg, seg, prune, lds, threads, stream); break;
    default: launch_mode<LJ_GENERIC>(A, energy, minimg, seg, prune, lds, threads, stream); break;
    }
}

}  // namespace mhip

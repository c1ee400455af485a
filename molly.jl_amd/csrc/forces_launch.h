// forces_launch.h — host-side entry points of the pair-kernel translation units (forces_inst.hip × 8, forces_uniform.hip)
#pragma once
#include "kernels.h"

namespace mhip {

// every k_forces<T, ·, COULM, ·, ·, ·, ·> variant, one explicit specialisation per translation unit
template <class T, int COULM>
void launch_forces_tc(const ForceArgs<T>& A, int ljm, bool energy, bool minimg, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream);

// the fp32 one-type LJ variants (forces only, block-local coordinates)
void launch_forces_uniform_f32(const ForceArgs<float>& A, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream);

}  // namespace mhip
